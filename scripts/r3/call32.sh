#!/bin/bash
# round 3, call 32: class-first sampling in fast mode -- the chi-square suites, then the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fast_unit_gpu.py tests/test_alias_trim_fast_gpu.py tests/test_edge_cases_gpu.py tests/test_scale_props_gpu.py tests/test_walk_gpu.py -q -m gpu > gpurun_out/r3af_tests_fast.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r3af_tests_fast.log
T0=$(date +%s); python bench.py > gpurun_out/r3af_bench_cfg4.json 2> gpurun_out/r3af_bench_cfg4.err
echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3af_bench_cfg4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")})
for k in ("biased", "fast_mode"):
    print(k, {kk: d[k][kk] for kk in d[k] if kk in ("value", "ms_per_step", "trials_per_step", "parity")})
    print("   roofline", {kk: d[k]["roofline"].get(kk) for kk in ("achieved", "frac", "kernel", "kernel_ms", "sectors_per_walk_step")})
PY
