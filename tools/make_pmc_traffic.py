#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: KiB.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B
request for wide (16 B/lane) loads, i.e. exactly half -- calibrated here on the geometry kernel of the depth-only
workloads, which reads the 48,000,000-byte triangle list once with dwordx4 loads -- so fetch bytes are doubled for the
streaming kernels (geometry, binning).  The tile kernel's reads are scattered 64-byte record gathers; that pattern was
calibrated with k_debug_gather (tools/pmc_gather_calib.py, profiles/r01_pmc: 8M distinct 64-B records = 524,288 KiB
true, FETCH_SIZE 523,779 KiB = 0.999x), so that part of its FETCH_SIZE is used as is; its reads of the (triangle, tile)
lists, however, are contiguous 16-byte-per-lane loads (the binning's records in the pre-pass, the near-first copy in the
main loop: 2 x 16 B per list entry), which count half like any wide streaming load: with the number of list entries of the
frame (6th argument, from frr_get_stats) the entry carries `fetch_list_correction_bytes` = 16 B x entries (the missing
half of 32 B x entries) and `hbm_bytes_per_launch` includes it -- an upper bound, since entries served by the segment
table walk of lightly loaded tiles are read once.  WRITE_SIZE is exact for full-line stores.
Every entry carries the digest of the kernel sources it was measured on (`_source_sha`): bench.py reports the
traffic only while the library it runs was built from the same sources.
With a third counter directory (a pass holding SQ_INSTS_VALU / SQ_INSTS_SALU) the tile kernel's entry also carries
`valu_issue`: the VALU / SALU instructions one launch issues and the mean issue cost of one of its VALU instructions
(tools/valu_mix.py: static mix of the kernel's ISA x the measured per-kind costs) -- what bench.py prices the VALU-issue
bound with.
usage: make_pmc_traffic.py <fetch_dir> <write_dir> <workload> <out.json> [bin_entries [sq_dir]]"""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "f_renderer_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")):
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def mean_counter(d, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fd, wd, workload, out = sys.argv[1:5]
    bin_entries = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    sq_dir = sys.argv[6] if len(sys.argv) > 6 else None
    valu = mean_counter(sq_dir, "SQ_INSTS_VALU") if sq_dir else {}
    salu = mean_counter(sq_dir, "SQ_INSTS_SALU") if sq_dir else {}
    fetch, write = mean_counter(fd, "FETCH_SIZE"), mean_counter(wd, "WRITE_SIZE")
    res = json.load(open(out)) if os.path.exists(out) else {}
    entry = {"_source_sha": source_sha()}
    calib = [v for k, v in fetch.items() if "k_geom_single<0>" in k or "k_geom_single<(int)0>" in k]
    if calib and "1M" in workload:
        entry["_calibration"] = {"kernel": "k_geom_single<0>", "fetch_size_kib": calib[0], "true_bytes": 48_000_000,
                                 "ratio": calib[0] * 1024 / 48_000_000}
    for k in fetch:
        if "frr::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "").replace("frr::", "")
        name = "k_raster" if short.startswith("k_raster_span") and "false" in short else short
        f_kib, w_kib = fetch[k], write.get(k, 0.0)
        corr = 1.0 if short.startswith("k_raster") else 2.0
        lists = 16 * bin_entries if short.startswith("k_raster") else 0
        entry[name] = {"kernel": short, "fetch_size_kib": round(f_kib, 1), "write_size_kib": round(w_kib, 1),
                       "fetch_correction": corr, "fetch_list_correction_bytes": lists,
                       "hbm_bytes_per_launch": round((corr * f_kib + w_kib) * 1024) + lists}
        if name == "k_raster" and k in valu:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import valu_mix
            m = re.match(r"k_raster_span<(\d+), (\d+), (true|false), (\d+), (\d+)>", short)
            key = "k_raster_spanILi%sELi%sELb%dELi%sELi%sEE" % (m.group(1), m.group(2), m.group(3) == "true", m.group(4), m.group(5))
            mix = valu_mix.mix(valu_mix.isa(), key)
            entry[name]["valu_issue"] = {"insts_valu": round(valu[k]), "insts_salu": round(salu.get(k, 0)), "cycles_per_valu": mix["cycles_per_valu"],
                                         "static_mix": {x: mix[x] for x in ("fast", "slow", "trans")}}
    res[workload] = entry
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(workload, json.dumps(entry.get("k_raster")))


if __name__ == "__main__":
    main()
