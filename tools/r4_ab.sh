#!/bin/bash
# round 4 dev: every library under tools/var/ on the named workloads, one frame at a time, REPS times each (interleaved)
mkdir -p gpurun_out
out=gpurun_out/r4_ab_${TAG:-x}.log; : > $out
for rep in $(seq 1 ${REPS:-2}); do
  for f in tools/var/libfrr_*.so; do
    n=$(basename $f .so); n=${n#libfrr_}
    FRR_LIB=$f TAG=$n FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 python tools/time_configs.py ${WL:-headline cfg4 cfg5} >> $out 2>&1 || echo "$n failed" >> $out
  done
done
sort -k2,2 -k1,1 $out
