#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: KiB.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B
request for wide (16 B/lane) loads, i.e. exactly half -- calibrated here on k_geom_count, which reads
the 48,000,000-byte triangle list once with dwordx4 loads (FETCH_SIZE reads 23,4xx KiB = 0.50x) -- so
fetch bytes are doubled for the streaming kernels (geometry, binning).  The tile kernel's reads are
scattered 64-byte record gathers; that pattern is calibrated separately with k_debug_gather
(tools/pmc_gather_calib.py: 8M distinct 64-B records = 524,288 KiB true, FETCH_SIZE 523,779 KiB =
0.999x), so its FETCH_SIZE is used as is.  WRITE_SIZE is exact (k_clear: 24,300 KiB for 3 x 8,294,400 B).
usage: make_pmc_traffic.py <fetch_dir> <write_dir> <workload> <out.json> [<gather_calib_dir>]"""
import csv, glob, json, os, sys
from collections import defaultdict


def mean_counter(d, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fd, wd, workload, out = sys.argv[1:5]
    fetch, write = mean_counter(fd, "FETCH_SIZE"), mean_counter(wd, "WRITE_SIZE")
    res = json.load(open(out)) if os.path.exists(out) else {}
    entry = {}
    calib = [v for k, v in fetch.items() if "k_geom_count" in k]
    if calib:
        entry["_calibration"] = {"kernel": "k_geom_count", "fetch_size_kib": calib[0], "true_bytes": 48_000_000,
                                 "ratio": calib[0] * 1024 / 48_000_000}
    if len(sys.argv) > 5:
        g = [v for k, v in mean_counter(sys.argv[5], "FETCH_SIZE").items() if "k_debug_gather" in k]
        if g:
            entry["_calibration_gather"] = {"kernel": "k_debug_gather", "fetch_size_kib": g[0], "true_bytes": 64 << 23,
                                            "ratio": g[0] * 1024 / (64 << 23)}
    for k in fetch:
        if "frr::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "").replace("frr::", "")
        name = "k_raster" if short.startswith("k_raster_span") and "false" in short else short
        f_kib, w_kib = fetch[k], write.get(k, 0.0)
        corr = 1.0 if short.startswith("k_raster") else 2.0
        entry[name] = {"kernel": short, "fetch_size_kib": round(f_kib, 1), "write_size_kib": round(w_kib, 1),
                       "fetch_correction": corr,
                       "hbm_bytes_per_launch": round((corr * f_kib + w_kib) * 1024)}
    res[workload] = entry
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry.get("k_raster"), indent=1))


if __name__ == "__main__":
    main()
