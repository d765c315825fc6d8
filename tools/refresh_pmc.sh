#!/bin/bash
# Dev tool, run on the GPU box: only the rocprofv3 passes profiles/pmc_traffic.json and profiles/r04_pmc/ are made from
# (kernel trace one frame at a time + the four counter passes per workload), into gpurun_out/final4/.  Then tools/collect_profiles.py.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final4; mkdir -p $O
export FRR_FRAMES_IN_FLIGHT=1 FRR_OVERLAP=0
for w in headline cfg4 cfg5; do
  rm -rf $O/kts_$w $O/fetch_$w $O/write_$w $O/sq1_$w $O/sq2_$w
  rocprofv3 --kernel-trace --stats -d $O/kts_$w --output-format csv -- python3 bench.py --workload $w --also none --steps 20 --warmup 3 --cpu-baseline-seconds 0 > $O/kts_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/f_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/w_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_ANY -d $O/sq1_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/s1_$w.log 2>&1 || exit 1
  FRAMES=4 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/sq2_$w --output-format csv -- python3 tools/pmc_frame.py $w > $O/s2_$w.log 2>&1 || exit 1
done
echo pmc refreshed
