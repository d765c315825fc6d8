#!/bin/bash
# round 4 dev: GPU suite, then times one frame at a time (TAG names the build)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_suite_${TAG:-x}.log 2>&1; rc=$?
tail -5 gpurun_out/r4_suite_${TAG:-x}.log
[ $rc -ne 0 ] && exit $rc
FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 python tools/time_configs.py headline cfg4 cfg5 cfg3 2>&1 | tee gpurun_out/r4_time_${TAG:-x}.log
