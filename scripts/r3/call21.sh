#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3y_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3y_tests.log
tail -4 gpurun_out/r3y_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3y_smoke.log 2>&1 || { tail -5 gpurun_out/r3y_smoke.log; exit 1; }
tail -1 gpurun_out/r3y_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r3y_bench_cfg4.json 2> gpurun_out/r3y_bench_cfg4.err
echo "bench rc=$?"; tail -c 300 gpurun_out/r3y_bench_cfg4.json
timeout -k 10 300 python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r3y_bench_cfg3.json 2> gpurun_out/r3y_bench_cfg3.err
echo "bench cfg3 rc=$?"
