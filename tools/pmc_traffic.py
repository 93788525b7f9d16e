#!/usr/bin/env python3
"""profiles/<tag>_pmc_*.csv -> profiles/pmc_traffic.json: HBM-side bytes per launch of every ldh:: kernel.

FETCH_SIZE / WRITE_SIZE are in KiB (checked: k_match_lazy writes 4 B x 2^30 positions and WRITE_SIZE says 4194324).
On gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled (MI355X_MICROARCH.md, HBM section; checked here too:
k_match_lazy must read V (1 GiB) + links (2 GiB) once, and 2 x FETCH_SIZE = 3.0 GiB)."""
import csv, json, sys
src, dst, workload = sys.argv[1], sys.argv[2], sys.argv[3]
per = {}
for r in csv.DictReader(open(src)):
    k = r["kernel"].replace("ldh::", "").split("<")[0]
    per.setdefault(k, {})[r["counter"]] = float(r["per_dispatch"])
out = {"workload": workload, "source": src, "unit": "bytes per launch", "kernels": {}}
for k, c in sorted(per.items()):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["kernels"][k] = {"read": c["FETCH_SIZE"] * 1024 * 2, "write": c["WRITE_SIZE"] * 1024,
                             "total": c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
