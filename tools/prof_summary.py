#!/usr/bin/env python3
"""Trim a rocprofv3 --stats kernel_stats.csv to this library's kernels (ldh::*) and print it as a
small CSV (what gets committed under profiles/)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["Name"] = r["Name"].replace("void ldh::", "ldh::")
tot = sum(int(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("ldh::"))
print("kernel,calls,total_ms,avg_ms,share_of_ldh_pct,min_ms,max_ms")
for r in rows:
    if r["Name"].startswith("ldh::"):
        t = int(r["TotalDurationNs"])
        print("%s,%s,%.3f,%.4f,%.2f,%.4f,%.4f" % (r["Name"].split("(")[0].replace(", ", ";"), r["Calls"], t / 1e6, float(r["AverageNs"]) / 1e6,
                                              100.0 * t / tot, int(r["MinNs"]) / 1e6, int(r["MaxNs"]) / 1e6))
