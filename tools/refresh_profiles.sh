#!/bin/bash
# Dev tool, run on the GPU box (gpurun): everything profiles/ is generated from, written under gpurun_out/final/.
#   gpurun -- 'bash tools/refresh_profiles.sh'   then   python tools/collect_profiles.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --cpu-baseline-seconds 0 > $O/kt.log 2>&1 || exit 1
FRAMES=4 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 tools/pmc_frame.py > $O/f.log 2>&1 || exit 1
FRAMES=4 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 tools/pmc_frame.py > $O/w.log 2>&1 || exit 1
FRAMES=4 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_ANY -d $O/sq1 --output-format csv -- python3 tools/pmc_frame.py > $O/s1.log 2>&1 || exit 1
FRAMES=4 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/sq2 --output-format csv -- python3 tools/pmc_frame.py > $O/s2.log 2>&1 || exit 1
timeout -k 10 600 python tools/run_configs.py --json $O/configs.json > $O/cfg.log 2>&1 || exit 1
for p in 0,1 0,2 0,4 0,8; do echo "PART=$p 1920x1080" >> $O/part.log; PART=$p python tools/quick_time.py 2>&1 | grep frame | cut -c1-60 >> $O/part.log; done
for p in 0,1 0,2 0,4 0,8; do echo "PART=$p 4096x4096" >> $O/part.log; PART=$p python tools/quick_time.py 4096 4096 1000000 2>&1 | grep frame | cut -c1-60 >> $O/part.log; done
cut -c1-200 $O/bench.json; cat $O/part.log
