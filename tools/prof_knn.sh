#!/bin/bash
# On the GPU box: kernel trace and one counter pass over tools/time_knn.py -> per-kernel time, vector
# instructions and pipe occupancy of distCUDA2.  usage: tools/prof_knn.sh [POINTS]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/knn_t /tmp/knn_c
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/knn_t -- python3 $R/tools/time_knn.py "$@" > /tmp/knn_t.log 2>&1 || { echo "trace failed"; tail -3 /tmp/knn_t.log; exit 1; }
grep distCUDA2 /tmp/knn_t.log
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/knn_c -- python3 $R/tools/time_knn.py "$@" > /tmp/knn_c.log 2>&1 || { echo "counter pass failed"; tail -3 /tmp/knn_c.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for row in csv.DictReader(open(glob.glob("/tmp/knn_t/**/*kernel_stats.csv", recursive=True)[0])):
    if "gs::" in row["Name"]:
        print("%-46s calls %3s avg_us %10.1f" % (row["Name"].split("(")[0][-46:], row["Calls"], float(row["AverageNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/knn_c/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "knn_search" in row["Kernel_Name"]:
            acc["knn_search"][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(d.items())})
PY
