#!/bin/bash
# usage: tools/ab_bench.sh NAME...  -> runs bench.py once per tools/lib_NAME.so (same box, back to back)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  GSPLAT_AMD_LIB=$PWD/tools/lib_$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'],1), {k:round(x['ms'],3) for k,x in d['kernels'].items() if x['ms']>0.2 or k=='host_sync_gap'})" || exit 1
done
