#!/bin/bash
# Counter totals of k_render for one library build: tools/pmc_ab.sh <variant> "<counters>" <bench args...>
# (variant = suffix of raytracing-one-weekend_amd/lib/libwrt_<suffix>.so, "base" = libwrt.so)
V=$1; shift
COUNTERS=${1:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$ROOT/raytracing-one-weekend_amd/lib/libwrt_$V.so
[ "$V" = base ] && lib=$ROOT/raytracing-one-weekend_amd/lib/libwrt.so
OUT=$ROOT/gpurun_out/pmcab_$V
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WRT_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc $COUNTERS --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/log.txt 2>&1 || { echo pmc failed; tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_render' in row['Kernel_Name']:
            tot[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
for k in sorted(tot): print("$V", k, "%.4g per dispatch"%(tot[k]/max(1,n[k])))
PY
