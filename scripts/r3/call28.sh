#!/bin/bash
# round 3, call 28: the end-to-end oracle test of fit_streaming, the larger oracle samples of the
# full-size configurations, then the whole GPU suite as the driver runs it
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_api_gpu.py -q -m gpu --durations=5 > gpurun_out/r3ac_api.log 2>&1
echo "api rc=$?"; tail -12 gpurun_out/r3ac_api.log
timeout -k 10 900 python -m pytest tests -q -m gpu -x --durations=12 > gpurun_out/r3ac_tests_gpu.log 2>&1
echo "suite rc=$?"; tail -20 gpurun_out/r3ac_tests_gpu.log
