#!/bin/bash
# Profiles one bench.py workload on the GPU box: kernel-trace statistics and
# two separate PMC passes (never combined with tracing).  Usage:
#   tools/profile.sh <tag> <bench args...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $OUT/pmc1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $OUT/pmc2.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/pmc3.log 2>&1 || { echo pmc3 failed; tail -5 $OUT/pmc3.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/pmc4.log 2>&1 || { echo pmc4 failed; tail -5 $OUT/pmc4.log; exit 1; }
find $OUT -name "*.csv" | head -20
python3 $ROOT/tools/summarise_profile.py $OUT
