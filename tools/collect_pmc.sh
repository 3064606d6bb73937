#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group over a short
# bench run, plus one kernel-trace pass.  Output under gpurun_out/pmc/.
# usage: tools/collect_pmc.sh [extra bench.py args]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc
rm -rf "$OUT"; mkdir -p "$OUT"   # (delete the local gpurun_out/pmc too before collecting: results are merged into it)
# one view after the other (--streams 1): counters and durations per kernel, undisturbed by other views' kernels
BENCH="bench.py --steps 2 --warmup 1 --streams 1 --no-extras --no-cpu-baseline $*"
run_pass() {
    name=$1; shift
    echo "== pass $name: $*"
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $BENCH > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run_pass fetch FETCH_SIZE &&
run_pass write WRITE_SIZE &&
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT &&
run_pass sq2 SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_UNALIGNED_STALL &&
run_pass tcc TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE &&
echo "== kernel trace" &&
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --streams 1 --no-extras --no-cpu-baseline $* > "$OUT/trace.log" 2>&1 &&
grep '^{"metric"' "$OUT/trace.log" > "$OUT/bench_under_rocprof.json" &&
echo "== kernel trace, default streams" &&
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_streams" -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline $* > "$OUT/trace_streams.log" 2>&1 &&
grep '^{"metric"' "$OUT/trace_streams.log" > "$OUT/bench_under_rocprof_streams.json" &&
find "$OUT" -name "*.csv" | head -30
