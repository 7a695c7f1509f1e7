"""Does page-locking host memory on a helper thread stall kernel launches of the main thread?  Main: a tight loop of tiny
launches, host timestamp per call; helper: torch pinned allocations (4 x 1.35 GB), hipHostRegister of a pageable buffer in
chunks, or plain first-touch.  Prints the largest gaps between consecutive launch calls and the loop's total time."""
import ctypes, json, sys, os, threading, time, torch
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
def loop(n=60000):
    ts = []
    t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1.0)
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    gaps = sorted((b - a for a, b in zip(ts, ts[1:])), reverse=True)[:5]
    return {"total_s": round(total, 3), "largest_gaps_ms": [round(g * 1e3, 2) for g in gaps]}
res = {"alone": loop()}
hip = ctypes.CDLL("libamdhip64.so")
def helper(kind):
    time.sleep(0.05)
    t0 = time.perf_counter()
    if kind == "torch_pin":
        helper.keep = [torch.empty((512, 10, 256, 256), dtype=torch.float32, pin_memory=True) for _ in range(4)]
    elif kind == "register_chunks":
        buf = torch.empty((4, 512, 10, 256, 256), dtype=torch.float32)
        chunk = buf.numel() * 4 // 32
        for i in range(32):
            rc = hip.hipHostRegister(ctypes.c_void_p(buf.data_ptr() + i * chunk), ctypes.c_size_t(chunk), 0)
            assert rc == 0, rc
        helper.keep = buf; helper.unreg = [(buf.data_ptr() + i * chunk) for i in range(32)]
    elif kind == "touch":
        helper.keep = torch.empty((4, 512, 10, 256, 256), dtype=torch.float32).fill_(1.0)
    helper.t = time.perf_counter() - t0
for kind in ("torch_pin", "register_chunks", "touch"):
    th = threading.Thread(target=helper, args=(kind,)); th.start()
    r = loop(); th.join(); r["helper_s"] = round(helper.t, 3)
    res[kind] = r
    if kind == "register_chunks":
        for p in helper.unreg: hip.hipHostUnregister(ctypes.c_void_p(p))
    helper.keep = None; torch._C._host_emptyCache()
print(json.dumps(res))
