"""Does a helper thread inside hipHostMalloc / hipHostRegister block hipMalloc (torch's allocator asking the driver for a new block)
on the main thread?  Main: loop of {torch.empty(64 MB, device) ; del ; empty_cache()} with a host timestamp per iteration."""
import ctypes, json, threading, time, torch
dev = torch.device("cuda", 0)
def loop(n=400):
    ts = [time.perf_counter()]
    for _ in range(n):
        t = torch.empty(64 << 20, dtype=torch.uint8, device=dev); del t; torch.cuda.empty_cache()
        ts.append(time.perf_counter())
    gaps = sorted((b - a for a, b in zip(ts, ts[1:])), reverse=True)[:5]
    return {"total_s": round(ts[-1] - ts[0], 3), "largest_ms": [round(g * 1e3, 2) for g in gaps]}
res = {"alone": loop()}
hip = ctypes.CDLL("libamdhip64.so")
def helper(kind):
    time.sleep(0.02); t0 = time.perf_counter()
    if kind == "torch_pin":
        helper.keep = [torch.empty((512, 10, 256, 256), dtype=torch.float32, pin_memory=True) for _ in range(4)]
    else:
        buf = torch.empty((4, 512, 10, 256, 256), dtype=torch.float32)
        nchunk = 32 if kind == "register_32" else 128
        chunk = buf.numel() * 4 // nchunk
        for i in range(nchunk):
            assert hip.hipHostRegister(ctypes.c_void_p(buf.data_ptr() + i * chunk), ctypes.c_size_t(chunk), 0) == 0
            if kind.endswith("sleep"): time.sleep(0.0005)
        helper.keep = buf; helper.unreg = [(buf.data_ptr() + i * chunk) for i in range(nchunk)]
    helper.t = time.perf_counter() - t0
for kind in ("torch_pin", "register_32", "register_128_sleep"):
    helper.unreg = []
    th = threading.Thread(target=helper, args=(kind,)); th.start()
    r = loop(); th.join(); r["helper_s"] = round(helper.t, 3); res[kind] = r
    for p in helper.unreg: hip.hipHostUnregister(ctypes.c_void_p(p))
    helper.keep = None; torch._C._host_emptyCache()
print(json.dumps(res))
