#!/bin/bash
# round 4 dev: baseline times one frame at a time + the binning chunk sweep on the 4096^2 frame
export FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0
mkdir -p gpurun_out
{
TAG=base python tools/time_configs.py headline cfg4 cfg5
for g in 64 96 128 192; do FRR_BIN_G=$g TAG=binG$g python tools/time_configs.py cfg4; done
for g in 128 192; do FRR_BIN_G=$g TAG=binG$g python tools/time_configs.py headline; done
} > gpurun_out/r4_base.log 2>&1
cat gpurun_out/r4_base.log
