#!/bin/bash
# usage: tools/ab_env.sh "VAR=1" "VAR=2 OTHER=x" ...  -> runs bench.py once per environment setting ("-" = none) with the in-tree library, same box, back to back
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in "$@"; do
  [ "$e" = "-" ] && e="GSR_AB_NONE=1"
  env $e timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$e]', round(d['value'],1), round(d.get('one_stream_it_s') or 0,1), {k:round(x['ms'],3) for k,x in d['kernels'].items() if x['ms']>0.05})" || exit 1
done
