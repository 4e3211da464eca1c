#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter from rocprofv3 --output-format csv counter_collection files."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names) + ",calls")
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur_ns"])):
    print('"' + k.replace('"', "'") + '",' + ",".join("%.6g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in names) + ",%d" % max(len(v) for v in d.values()))
