"""Build profiles/traffic.json (HBM bytes per launch of the step kernels) from a tests/prof.sh output directory.
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 on gfx950 tallies the 128-byte requests of 16-byte/lane
streams as 64 B (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact (calibrated on the pure-streaming k_dwdt of
the earlier builds: 3*S measured to 0.002 %, profiles/r01_final_*)."""
import json
import re
import sys

src, n, B, dtype = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
nchunks = int(sys.argv[5]) if len(sys.argv) > 5 else 1   # launches per full-batch pass (cache-sized chunks, bench.py prints it)
txt = open(f"{src}/summary.txt").read()
blocks = re.split(r"\n(?=k_)", txt.split("== PMC (mean per dispatch)\n")[1])
S = B * n * (n // 2 + 1) * (16 if dtype == "f64" else 8)
out = {}
for b in blocks:
    lines = b.strip().split("\n")
    name = lines[0].strip()
    vals = {l.split()[0]: float(l.split()[1]) for l in lines[1:] if len(l.split()) >= 2}
    m = re.match(r"k_cols<\w+, (\d+), \d+, \d+, (\d)", name)
    if m and int(m.group(1)) == n:
        key, alg = {0: ("k_cols<MODE_A>", 5), 1: ("k_cols<MODE_CA>", 9), 2: ("k_cols<MODE_C>+dwdt", 7)}.get(int(m.group(2)), (None, 0))
    elif name.startswith("k_rows_advect") and (f", {n}," in name or (name.startswith("k_rows_advect7") and n == 1024)):
        key, alg = "k_rows_advect", 5
    elif name.startswith("k_dwdt"):
        key, alg = "k_dwdt", 3
    else:
        key = None
    if not key or "FETCH_SIZE" not in vals:
        continue
    traffic = (2 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0)) * 1024 * nchunks   # per PASS over the whole batch
    out[f"{key}|n{n}|B{B}|{dtype}"] = round(traffic)
    print(f"{key:18s} algorithmic {alg*S/1e9:6.2f} GB   measured {traffic/1e9:6.2f} GB   (TCC_MISS*128 = {vals.get('TCC_MISS_sum',0)*128/1e9:.2f} GB)")
out["_note"] = ("bytes between L2 and the memory side per full-batch PASS of a kernel (= launches_per_pass chunk launches) = "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch x launches per pass, from separate rocprofv3 --pmc passes "
                "(tests/prof.sh -> profiles/*_rocprofv3_summary.txt); Infinity-Cache hits are included in these counters")
out["_build"] = sys.argv[6] if len(sys.argv) > 6 else "unlabelled"
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
