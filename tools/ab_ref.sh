#!/bin/bash
# usage: tools/ab_ref.sh NAME...  -> bench.py --api reference once per tools/lib_NAME.so ("main" = the in-tree library), twice, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/tools/lib_$v.so; [ "$v" = main ] && lib=$PWD/gaussian-splatting_amd/libgsplat_amd.so
  GSPLAT_AMD_LIB=$lib timeout -k 10 200 python bench.py --api reference --steps 6 --warmup 2 --no-extras --no-cpu-baseline --streams 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), {k:round(x['ms'],3) for k,x in d['kernels'].items() if x['ms']>0.1})" || exit 1
done; done
