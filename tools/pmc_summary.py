#!/usr/bin/env python3
"""Average PMC counter values per kernel from a rocprofv3 --pmc csv run."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
