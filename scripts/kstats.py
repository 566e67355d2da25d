"""per-query kernel table from a rocprofv3 kernel_stats csv: python scripts/kstats.py <csv> <queries_run>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0
for r in rows:
    if "read_reduce" in r["Name"]:
        continue
    t = int(r["TotalDurationNs"]); tot += t
    name = r["Name"][:64]
    print(f"{name:64s} calls/q {int(r['Calls'])/steps:5.1f} us/q {t/steps/1e3:7.1f}")
print("total us/query", round(tot / steps / 1e3, 1))
