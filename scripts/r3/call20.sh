#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py -m gpu -q > gpurun_out/r3w_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3w_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-baseline --no-sgns --no-regimes > gpurun_out/r3w_bench_walks_cfg4.json 2> gpurun_out/r3w_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3w_bench_walks_cfg4.json') if l.startswith('{')][-1])
print('value %.4g'%d['value'], 'ms', d['ms_per_step']); r=d['roofline']; print({k:r[k] for k in ('frac','hop_table','algorithmic_bytes_per_walk_step','kernel_ms')}); print(r['gather_ceiling']); print(d['setup'])
PY
