#!/bin/bash
# round 6: the final measurements of the round in one call -- the GPU suite, the default bench line, the sequential callers, the real-batch kernel sequence
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -s --durations=10 > gpurun_out/gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_suite.log; tail -4 gpurun_out/gpu_suite.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err || echo "bench failed"
timeout -k 10 300 python bench.py --workload odometry --steps 40 > gpurun_out/r06_bench_odometry.json 2> gpurun_out/r06_bench_odometry.err || echo "odometry bench failed"
timeout -k 10 300 python bench.py --workload mapmaker --steps 40 > gpurun_out/r06_bench_mapmaker.json 2> gpurun_out/r06_bench_mapmaker.err || echo "mapmaker bench failed"
bash scripts/kseq.sh --workload sample > gpurun_out/r06_kseq_sample.txt 2>&1
bash scripts/kseq.sh > gpurun_out/r06_kseq_default.txt 2>&1
python - <<'PY'
import json
for f in ("r06_bench_default", "r06_bench_odometry", "r06_bench_mapmaker"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().split("\n")[-1])
        print(f, d["value"], d["unit"], d.get("ms_per_step"), {k: d["roofline"].get(k) for k in ("frac", "whole_path_frac", "avg_launch_ms", "keyframe_ms_per_step", "gn_loop_ms_per_step")}, d.get("wall_s_by_phase"), (d.get("burst") or {}).get("frames_per_s"))
        for k in ("latency", "highres", "sample_batch", "other_storage_order", "ctor"):
            if d.get(k): print("   ", k, {kk: vv for kk, vv in d[k].items() if kk in ("ms_per_pair", "pairs_per_s", "ms_per_step", "slowdown_vs_synthetic_at_equal_point_count", "roofline_frac", "keyframe_ms_per_step", "gn_loop_ms_per_step", "adapter_class_ICET_ms", "max_abs_dX_vs_oracle_first_4_pairs", "max_abs_dX_vs_oracle")})
    except Exception as e:
        print(f, "unreadable:", e)
PY
