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
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    res[f"{run} {counter}"] = [{"kernel": k, "counter": counter, "dispatches": len(v), "avg": sum(v) / len(v)}
                               for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))]
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, "runs:", list(res.keys()))
