"""collects rocprofv3 counter_collection csvs of scripts/pmc_r2.sh into one JSON:
{"<run> <COUNTER>": [{"kernel", "counter", "dispatches", "avg"}...]} (per-dispatch averages, KiB)"""
import collections, csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
res = {}
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    if not os.path.isdir(d):
        continue
    run, counter = os.path.basename(d).rsplit("_", 1)[0], os.path.basename(d).split("_", 1)[1]
    run = os.path.basename(d)[: -len(counter) - 1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    # a kernel launched over several shapes in one run (ph_gather_multi over 3.3 M and over 0.4 M rows) is
    # reported per launch shape (grid size); the entry of a kernel is its shape with the most traffic
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]][row.get("Grid_Size", "")].append(float(row["Counter_Value"]))
    entries = []
    for k, shapes in acc.items():
        g, v = max(shapes.items(), key=lambda kv: sum(kv[1]))
        entries.append({"kernel": k, "counter": counter, "dispatches": len(v), "avg": sum(v) / len(v), "grid_size": g,
                        "other_shapes": {gg: {"dispatches": len(vv), "avg": sum(vv) / len(vv)} for gg, vv in shapes.items() if gg != g}})
    res[f"{run} {counter}"] = sorted(entries, key=lambda e: -e["avg"] * e["dispatches"])
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, "runs:", list(res.keys()))
