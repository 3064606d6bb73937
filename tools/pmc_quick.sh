#!/bin/bash
# Quick counter look at one kernel on the GPU box: FETCH_SIZE, WRITE_SIZE and one SQ group over a
# 3-step bench run, one rocprofv3 run per group.  usage: tools/pmc_quick.sh KERNEL_SUBSTRING [bench args]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
KERN=$1; shift
OUT=gpurun_out/pmc2
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline $*"
for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  set -- $grp; name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $BENCH > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
done
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for path in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
