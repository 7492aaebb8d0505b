#!/bin/bash
# round 3 call 1: full GPU suite (parity first), hogwild AUC distribution, default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
timeout -k 10 400 python scripts/r3/hogwild_auc_runs.py 20 > gpurun_out/r3a_hogwild_auc_runs.log 2>&1
tail -6 gpurun_out/r3a_hogwild_auc_runs.log
timeout -k 10 600 python bench.py > gpurun_out/r3a_bench_cfg4.json 2> gpurun_out/r3a_bench_cfg4.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r3a_bench_cfg4.json
