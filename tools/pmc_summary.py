#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection) per kernel: mean counter value per
dispatch.  Usage: python tools/pmc_summary.py gpurun_out/pmc1 [gpurun_out/pmc2 ...] [--json out.json]

HBM traffic (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads (x2 correction applies
to that access pattern only); WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = None
    if "--json" in sys.argv:
        out_json = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != out_json]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                key = (row.get("Dispatch_Id"), k)
                if key not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    seen.add(key)
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    res = {}
    for k in sorted(acc):
        if not k.startswith("frr::"):
            continue
        res[k] = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
        res[k]["_dispatches"] = max(len(v) for v in acc[k].values())
        if dur[k]:
            res[k]["_avg_us_profiled"] = sum(dur[k]) / len(dur[k])
    for k, v in res.items():
        print(k)
        for c, x in v.items():
            print(f"    {c:28s} {x:16.1f}")
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
