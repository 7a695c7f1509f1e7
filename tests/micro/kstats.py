"""Print selected rows of a rocprofv3 kernel_stats.csv: python tests/micro/kstats.py file.csv [name-substring ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print(f"all kernels: {tot:.2f} ms")
for r in rows:
    if len(sys.argv) < 3 or any(s in r["Name"] for s in sys.argv[2:]):
        print(f'{r["Name"][:64]:64s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us avg {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
