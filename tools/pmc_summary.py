#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per (kernel, counter) for ldh:: kernels."""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ";")  # (no commas inside a CSV field)
        if not k.startswith("ldh::"): continue
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
print("kernel,counter,sum,dispatches,per_dispatch")
for (k, c) in sorted(acc):
    print("%s,%s,%.0f,%d,%.0f" % (k, c, acc[(k, c)], n[(k, c)], acc[(k, c)] / n[(k, c)]))
