"""Per-kernel table of the SFNO config-5 forward + loss from a tests/prof_sfno.sh output directory: launches per forward, average
duration (kernel trace), algorithmic bytes per launch, bytes between L2 and the memory side from the PMC
passes -- (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the guide's gfx950 correction: 128-byte requests of 16-byte-per-lane streams are
tallied at 64 B -- and the L2 hit rate.  A kernel that is launched with several grid sizes (the output convolution works on ONE
channel) is reported at its LARGEST grid (the hidden layers).  Writes <dir>/sfno_traffic.json (bench.py reads
profiles/sfno_traffic.json)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
b, C, X, Y, T, mx, my, mt = 32, 10, 256, 256, 10, 24, 24, 5
A_H = b * C * X * Y * T * 4
A_1 = A_H // C
W = b * C * X * 2 * my * mt * 8            # (b, C, X, Q) half-transformed planes
V = b * C * 2 * mx * 2 * my * mt * 8       # truncated spectrum
LP = b * T * X * 144 * 8                   # one half-spectrum plane set of the loss (pitch 144)
ALGO = {  # kernel-name prefix -> (algorithmic bytes per launch, what moves)
    "k_pointwise<10, 40, 10": (3 * A_H, "read conv output + layer input, write activation"),
    "k_fwd_ty2": (A_H + W, "read activation, write (b,C,X,Q) planes"),
    "k_inv_ty2": (W + A_H, "read planes, write activation"),
    "k_x<float, 256, 8, 16, true": (W + V, "read planes, write kept kx"),
    "k_x<float, 256, 8, 16, false": (V + W, "read kept kx, write planes"),
    "k_contract_mfma": (2 * V + 4 * C * C * mx * my * mt * 8, "spectrum in / out + the four weight blocks"),
    "k_contract_lanes": (2 * V + 4 * C * C * mx * my * mt * 8, "spectrum in / out + the four weight blocks"),
    # (inference forms the projection's LAST time slice only -- LiftingOperator._through_the_spectrum; the full projection, A_1 + A_H,
    #  runs in training)
    "k_pointwise<10, 10, 10": ((A_1 + A_H) // T, "lifting projection, last time slice: one-channel input (+ L2-resident table) -> (b, C, X, Y, 1)"),
    "k_lift_spectrum": (V // C + V + (C + 1) * (V // (b * C)), "kept modes of the one-channel input + table modes -> kept modes of the projection"),
    "k_pointwise<10, 10, 1": (A_H + A_1, "channel reduction"),
    "k_loss_rows": (2 * A_1 + 2 * LP, "read x, y; write two half-spectrum plane sets"),
    "k_loss_cols": (2 * LP, "read the plane sets"),
}


def short(name):
    m = re.match(r"void (k_\w+)<(.*)>\(", name)
    return f"{m.group(1)}<{m.group(2)[:34]}>" if m else name[:60]


def is_ours(k):
    return k.startswith("k_")


# durations per (kernel, grid) from the kernel trace
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if is_ours(k):
            dur[(k, int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if is_ours(k):
            pmc[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = sorted({k for k, _ in dur})
table = {}
n_fwd = None
print(f"{'kernel (largest grid)':52s} {'launches':>8s} {'avg us':>9s} {'algo MB':>9s} {'algo TB/s':>9s} {'L2<->mem MB':>11s} {'ratio':>6s} {'L2 hit':>6s} {'LDS confl':>9s}")
for k in kernels:
    grids = sorted((g for kk, g in dur if kk == k), reverse=True)
    g = grids[0]
    d = dur[(k, g)]
    avg = sum(d) / len(d)
    algo = next((v for p, v in ALGO.items() if k.startswith(p)), None)
    # the PMC csv's Grid_Size is in work-items of x; match the largest
    pg = sorted((gg for kk, gg in pmc if kk == k), reverse=True)
    c = pmc[(k, pg[0])] if pg else {}
    mean = lambda name: (sum(c[name]) / len(c[name])) if name in c else None
    fetch, write = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    traffic = (2 * fetch + (write or 0)) * 1024 if fetch is not None else None
    hit, miss = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
    confl, ldsact = mean("SQ_LDS_BANK_CONFLICT"), mean("SQ_LDS_IDX_ACTIVE")
    ent = {"launches_in_profile": len(d), "avg_us": round(avg, 1), "grid": g}
    if algo:
        ent.update(algo_bytes=algo[0], algo_TBps=round(algo[0] / avg / 1e6, 2), what=algo[1])
    if traffic is not None:
        ent.update(traffic_bytes=round(traffic), fetch_KB=round(fetch), write_KB=round(write or 0))
    if hit is not None and miss is not None and hit + miss > 0:
        ent["l2_hit"] = round(hit / (hit + miss), 3)
    if confl is not None and ldsact:
        ent["lds_conflict_share"] = round(confl / ldsact, 3)
    table[k] = ent
    print(f"{k:52s} {len(d):8d} {avg:9.1f} {(algo[0] / 1e6 if algo else float('nan')):9.1f} {(algo[0] / avg / 1e6 if algo else float('nan')):9.2f} "
          f"{(traffic / 1e6 if traffic is not None else float('nan')):11.1f} {(traffic / algo[0] if (traffic is not None and algo) else float('nan')):6.2f} "
          f"{ent.get('l2_hit', float('nan')):6.3f} {ent.get('lds_conflict_share', float('nan')):9.3f}")
table["_note"] = ("per launch at the kernel's largest grid (hidden layers); traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 "
                  "--pmc passes of tests/bench_sfno.py (TRAIN=0): bytes between L2 and the memory side, Infinity-Cache hits included; the "
                  "activations (839 MB) exceed the 256 MB Infinity Cache, so this is close to DRAM traffic")
json.dump(table, open(os.path.join(out, "sfno_traffic.json"), "w"), indent=1)
