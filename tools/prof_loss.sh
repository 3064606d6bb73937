#!/bin/bash
# usage (on the GPU box): tools/prof_loss.sh NAME...  -> rocprofv3 kernel stats of tools/time_loss.py per tools/lib_NAME.so ("main" = in-tree)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  lib=$R/tools/lib_$v.so; [ "$v" = main ] && lib=$R/gaussian-splatting_amd/libgsplat_amd.so
  export GSPLAT_AMD_LIB=$lib
  rm -rf /tmp/p_$v
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -- python3 $R/tools/time_loss.py > /tmp/p_$v.log 2>&1 || { echo "failed $v"; tail -3 /tmp/p_$v.log; exit 1; }
  f=$(find /tmp/p_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[2])):
    n = row["Name"]
    if "gs::" in n:
        print(sys.argv[1], n.split("(")[0][-40:], row["Calls"], "avg_us", round(float(row["AverageNs"]) / 1e3, 2))
PY
done
