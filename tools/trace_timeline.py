"""Dev tool: per-kernel start/end times (us) of the last frames of a rocprofv3 --kernel-trace CSV.
  python tools/trace_timeline.py <kernel_trace.csv> [frames]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
short = lambda n: ("tile" if "k_raster" in n else "geom" if "k_geom_single" in n else "clip" if "k_geom_clip" in n else "bin" if "k_bin_seg" in n else n[:12])
tiles = [k for k in ks if "k_raster" in k[2]]
t0 = tiles[-nf - 1][0] if len(tiles) > nf else ks[0][0]
for s, e, n, q in ks:
    if s >= t0:
        print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:6.1f} us)  q{q:>3}  {short(n)}")
if len(tiles) > nf:
    span = (tiles[-1][1] - tiles[-nf - 1][1]) / nf / 1e3
    print(f"tile kernel end-to-end period over the last {nf} frames: {span:.1f} us")
