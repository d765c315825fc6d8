#!/bin/bash
# Dev tool, run on the GPU box (gpurun): everything profiles/r04_* is generated from, written under gpurun_out/final4/.
#   gpurun -- 'bash tools/refresh_profiles.sh'   then   python tools/collect_profiles.py
# rocprofv3: the program itself follows `--` (python3 ...), kernel trace only next to --pmc (separate passes per counter set).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final4; [ -z "$PART2" ] && rm -rf $O; mkdir -p $O
if [ -z "$PART2" ]; then
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 python bench.py --cpu-baseline-seconds 0 > $O/bench_serial.json 2>> $O/bench.err || exit 1   # one frame at a time, one stream (round 2's mode)
# kernel trace + stats of the default command (two frames in flight); the counter passes, workgroup shapes and tile timelines
# look at kernels one frame at a time on one stream (FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0): comparable with round 2
for w in headline cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$w --output-format csv -- python3 bench.py --workload $w --also none --steps 20 --warmup 3 --cpu-baseline-seconds 0 > $O/kt_$w.log 2>&1 || exit 1
  export FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0
  FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $O/kts_$w --output-format csv -- python3 bench.py --workload $w --also none --steps 20 --warmup 3 --cpu-baseline-seconds 0 > $O/kts_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/f_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/w_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_ANY -d $O/sq1_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/s1_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/sq2_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/s2_$w.log 2>&1 || exit 1
  unset FRR_FRAMES_IN_FLIGHT FRR_OVERLAP
done
fi
rm -f $O/overlap_modes.txt
for w in headline cfg4 cfg5; do for p in "" 3,8; do CFG=$w PART=$p N=100 python tools/overlap_probe.py 2>&1 | grep "overlap=" >> $O/overlap_modes.txt || exit 1; done; done
OVERLAP=2 FIF=2 CFG=headline N=40 rocprofv3 --kernel-trace -d $O/kt_overlap -o kt --output-format csv -- python3 tools/overlap_probe.py > $O/kt_overlap.log 2>&1 || exit 1
python tools/trace_timeline.py $O/kt_overlap/kt_kernel_trace.csv 4 > $O/frames_in_flight_kernel_timeline.txt || exit 1
timeout -k 10 600 python tools/run_configs.py --json $O/configs.json > $O/cfg.log 2>&1 || exit 1
GPU_MAX_HW_QUEUES=8 python tools/partition_times.py --json $O/partition_times.json > $O/part.log 2>&1 || exit 1
SERIAL=1 python tools/partition_times.py --json $O/partition_times_serial.json > $O/part_serial.log 2>&1 || exit 1
FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 python tools/exp_shapes.py headline cfg4 cfg5 > $O/shapes.log 2>&1 || exit 1
for w in headline cfg4 cfg5; do FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0 FRR_LIB=tools/libfrr_dbg.so FRR_DEBUG_TILES=1 FRR_DEBUG_PRINT=1 python tools/tile_timeline.py $w > $O/timeline_$w.log 2>&1 || exit 1; done
cut -c1-300 $O/bench.json; tail -3 $O/part.log
