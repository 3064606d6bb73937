"""Condenses gpurun_out/pmc/ (written by tools/collect_pmc.sh) into
profiles/<tag>_pmc_summary.csv, profiles/<tag>_kernel_stats.csv and
profiles/pmc_traffic.json (HBM bytes per launch per kernel, FETCH_SIZE doubled
as measured for wide coalesced reads on gfx950, WRITE_SIZE as is; both are
reported in KiB by rocprofv3).

usage: python tools/pmc_summary.py TAG [TRAFFIC_JSON]
TRAFFIC_JSON (default pmc_traffic.json, which bench.py reads for the default workload) names the
traffic file under profiles/; give another name for a collection on another scene.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "pmc")
KERNEL_KEYS = {  # kernel symbol substring -> bench.py kernel name
    "depth_keys_batch_kernel": "depth_order_batch:keys", "sort_hist_batch_kernel": "depth_order_batch:hist",
    "sort_scan_hist_batch_kernel": "depth_order_batch:scan", "onesweep_batch_kernel": "depth_order_batch:onesweep",
    "preprocess_fwd_kernel": "preprocess", "depth_keys_kernel": "depth_order:keys",
    "emit_keys_kernel": "emit_keys", "onesweep_kernel": "sort:onesweep", "sort_hist_kernel": "sort:hist",
    "sort_scan_hist_kernel": "sort:scan", "tile_ranges_kernel": "tile_ranges", "render_fwd_kernel": "render_fwd",
    "render_bwd_kernel": "render_bwd", "preprocess_bwd_kernel": "preprocess_bwd",
    "preprocess_bwd_batch_kernel": "preprocess_bwd_batch", "preprocess_fwd_batch_kernel": "preprocess_batch", "order_gather_kernel": "depth_order:gather", "ssim_fwd": "ssim_fwd",
    "ssim_bwd": "ssim_bwd", "scan_blocks_kernel": "scan", "finish_gstat_kernel": "finish_stats", "clear_reached_kernel": "bwd_clear", "tile_count_kernel": "tile_count:count", "tile_finish_kernel": "tile_count:finish", "emit_partition_kernel": "emit_keys", "chunk_map_kernel": "emit:chunk_map"}


def short(name, grid=0):
    """Logical kernel name; the two sorts (Gaussians by depth, instances by tile) run the same
    kernels and are told apart by their grid size."""
    for k, v in KERNEL_KEYS.items():
        if k in name:
            if v in ("sort:onesweep", "sort:hist"):
                big = grid > (2_000_000 if v == "sort:onesweep" else 400_000)
                return v if big else v.replace("sort:", "depth_order:")
            return v
    return None


# bench.py kernel name -> (detail name, launches per view) that make it up
COMPOSED = {
    # "sort:hist" only when the tile grid is too large for the rectangle-count path (then "tile_count" is absent)
    "sort": (("sort:hist", 1), ("sort:scan", 1), ("sort:onesweep", 1)),
    "tile_count": (("tile_count:count", 1), ("tile_count:finish", 1)),
    "depth_order": (("depth_order:keys", 1), ("depth_order:hist", 1), ("depth_order:onesweep", 4)),
    "emit_keys": (("depth_order:gather", 1), ("emit_keys", 1)),
    # per step: the depth orders of all its views in shared launches
    "depth_order_batch": (("depth_order_batch:keys", 1), ("depth_order_batch:hist", 1), ("depth_order_batch:scan", 1),
                          ("depth_order_batch:onesweep", 3)),   # 27-bit keys: three passes (four if a depth reaches 13107.2)
}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    traffic_json = sys.argv[2] if len(sys.argv) > 2 else "pmc_traffic.json"
    acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values per dispatch
    newest = {}   # one file per pass: the latest (earlier collections may have left theirs behind)
    for path in glob.glob(os.path.join(SRC, "*", "**", "*counter_collection.csv"), recursive=True):
        key = os.path.relpath(path, SRC).split(os.sep)[0]
        if key not in newest or os.path.getmtime(path) > os.path.getmtime(newest[key]):
            newest[key] = path
    for path in newest.values():
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"], int(row.get("Grid_Size") or 0))
                if k:
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + counters)
        for k in sorted(acc):
            n = max(len(v) for v in acc[k].values())
            w.writerow([k, n] + [("%.6g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else "") for c in counters])
    print("wrote", out)
    # HBM traffic per launch, merged per bench kernel name (sub-kernels summed per logical step)
    traffic = defaultdict(float)
    for k in acc:
        fs = acc[k].get("FETCH_SIZE")
        ws = acc[k].get("WRITE_SIZE")
        if not fs or not ws:
            continue
        per_launch = (2.0 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024.0
        traffic[k] = per_launch
    with open(os.path.join(ROOT, "profiles", traffic_json.replace(".json", "_detail.json")), "w") as f:
        json.dump({k: traffic[k] for k in sorted(traffic)}, f, indent=1)
    simple = {k: v for k, v in traffic.items() if ":" not in k}
    for name, parts in COMPOSED.items():
        if all(d in traffic for d, _ in parts if d not in ("sort:scan", "sort:hist")):
            simple[name] = sum(traffic.get(d, 0.0) * n for d, n in parts)
    out_d = {k: simple[k] for k in sorted(simple)}
    # vector-instruction counts and the shader clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs) for bench.py's
    # issue-rate roofline; durations come from the trace pass (kernel_stats, ns)
    dur_ns = {}
    for path in sorted(glob.glob(os.path.join(SRC, "trace", "**", "*kernel_stats.csv"), recursive=True),
                       key=os.path.getmtime)[-1:]:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Name"])
                if k and k not in ("sort:onesweep", "sort:hist"):
                    dur_ns[k] = float(row["AverageNs"])
    valu = {}
    for k in acc:
        iv = acc[k].get("SQ_INSTS_VALU")
        if not iv or ":" in k:
            continue
        ent = {"valu_insts": sum(iv) / len(iv)}
        la = acc[k].get("SQ_INSTS_LDS_ATOMIC")
        if la:
            ent["lds_atomic_insts"] = sum(la) / len(la)
        ga = acc[k].get("GRBM_GUI_ACTIVE")
        if ga and k in dur_ns:
            ent["clock_GHz"] = (sum(ga) / len(ga)) / 8.0 / dur_ns[k]
        # SQ_ACTIVE_INST_VALU counts, summed over the SIMDs, the time a vector instruction occupies its SIMD, in
        # units of four cycles; with the busy cycles of the launch (GRBM_GUI_ACTIVE over the 8 XCDs) it says how
        # full the vector pipes were, whatever the instruction mix costs per instruction
        av = acc[k].get("SQ_ACTIVE_INST_VALU")
        if av and ga:
            ent["valu_active_simd_cycles"] = 4.0 * sum(av) / len(av)
            ent["launch_cycles"] = (sum(ga) / len(ga)) / 8.0
        valu[k] = ent
    out_d["_valu"] = valu
    # bench.py defaults, which tools/collect_pmc.sh runs; another scene's file is never bench.py's default one
    out_d["_workload"] = [6000000, 1920, 1080]
    out_d["_scene"] = "street" if traffic_json == "pmc_traffic.json" else traffic_json[len("pmc_traffic_"):-len(".json")]
    # the sources the profiled library was built from: bench.py compares this with the sources it runs on
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
    from _gsplat_lib import source_hash
    out_d["_build"] = source_hash()
    out_d["_note"] = "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB, mean over the launches of the PMC passes"
    with open(os.path.join(ROOT, "profiles", traffic_json), "w") as f:
        json.dump(out_d, f, indent=1)
    print("wrote profiles/" + traffic_json, simple)
    # kernel stats of the trace pass
    stats = sorted(glob.glob(os.path.join(SRC, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    for path in stats[-1:]:
        dst = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
        with open(path) as f, open(dst, "w") as g:
            g.write(f.read())
        print("wrote", dst)
    stats = sorted(glob.glob(os.path.join(SRC, "trace_streams", "**", "*kernel_stats.csv"), recursive=True),
                   key=os.path.getmtime)
    for path in stats[-1:]:
        dst = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_streams.csv")
        with open(path) as f, open(dst, "w") as g:
            g.write(f.read())
        print("wrote", dst)
    for name in ("bench_under_rocprof", "bench_under_rocprof_streams"):
        j = os.path.join(SRC, name + ".json")
        if os.path.exists(j):
            with open(j) as f, open(os.path.join(ROOT, "profiles", f"{tag}_{name}.json"), "w") as g:
                g.write(f.read())


if __name__ == "__main__":
    main()
