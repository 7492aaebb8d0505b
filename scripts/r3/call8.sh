#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r3i_bench_cfg4.json 2> gpurun_out/r3i_bench_cfg4.err
echo "bench rc=$?"
tail -c 400 gpurun_out/r3i_bench_cfg4.json
timeout -k 10 700 python scripts/r3/e2e_cfg4.py batched 1.0 > gpurun_out/r3i_e2e_cfg4_batched.json 2> gpurun_out/r3i_e2e_cfg4_batched.err
echo "e2e rc=$?"
cat gpurun_out/r3i_e2e_cfg4_batched.json
