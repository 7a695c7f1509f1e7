"""Read a rocprofv3 kernel-trace CSV and print, for the LAST `count` launches, name / duration / gap to the previous kernel's end
(negative = overlap).  usage: trace_gaps.py <dir> [count]"""
import csv, glob, sys
d = sys.argv[1]; count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
prev_end = None; t0 = int(rows[0]["Start_Timestamp"])
tot_k = 0; tot_gap = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  grid {r.get('Grid_Size_X', '?'):>8} wg {r.get('Workgroup_Size_X', '?'):>4}  {r['Kernel_Name'][:90]}")
    tot_k += (e - s) / 1e3; tot_gap += max(gap, 0.0) if prev_end is not None else 0
    prev_end = max(prev_end, e) if prev_end is not None else e
print(f"span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {tot_k:.1f} us, sum of positive gaps {tot_gap:.1f} us")
