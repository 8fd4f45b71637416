#!/bin/bash
# A/B of library builds inside one GPU-box session: tools/ab.sh "<variants>" <bench args>
# a variant is a suffix of raytracing-one-weekend_amd/lib/libwrt_<suffix>.so ("base" = libwrt.so)
VARIANTS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for v in $VARIANTS; do
  lib=$ROOT/raytracing-one-weekend_amd/lib/libwrt_$v.so
  [ "$v" = base ] && lib=$ROOT/raytracing-one-weekend_amd/lib/libwrt.so
  WRT_LIB=$lib timeout -k 10 200 python3 -u $ROOT/bench.py --no-cpu "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v round $round', d['config']['workload'][:6], 'Mrays/s', d['value'], 'ms', d['ms_per_step'], 'kernel_ms', r['kernel_ms_avg'], 'frac', r['frac'])" || exit 1
done
done
