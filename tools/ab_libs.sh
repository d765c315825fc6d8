#!/bin/bash
# Dev tool: bench.py with each library variant under tools/var/ -- one frame at a time (kernel times compare), or with
# AB_MODE=default the default mode (two frames in flight).
mkdir -p gpurun_out/ab
for f in tools/var/libfrr_*.so; do
  n=$(basename $f .so); n=${n#libfrr_}
  if [ "$AB_MODE" = "default" ]; then
    FRR_LIB=$f python bench.py --cpu-baseline-seconds 0 > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err || echo "$n failed"
  else
    FRR_LIB=$f FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 python bench.py --cpu-baseline-seconds 0 > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err || echo "$n failed"
  fi
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]; d=json.load(open(f"gpurun_out/ab/{n}.json"))
print("%-8s headline %.4f ms golden=%s |"%(n,d["ms_per_step"],d.get("image_matches_golden")), " | ".join("%s %.4f ms golden=%s"%(s["config"]["workload"][:6],s["ms_per_step"],s.get("image_matches_golden")) for s in d["secondary"]), flush=True)
PY
done
