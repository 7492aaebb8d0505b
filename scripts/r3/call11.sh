#!/bin/bash
# the driver's sequence on the current build + the round's profile of bench.py
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3l_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3l_tests.log
tail -4 gpurun_out/r3l_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3l_smoke.log 2>&1 || { tail -5 gpurun_out/r3l_smoke.log; exit 1; }
tail -1 gpurun_out/r3l_smoke.log
timeout -k 10 1000 bash scripts/r3/profile_r3.sh r3l_cfg4 > gpurun_out/r3l_profile.log 2>&1
tail -3 gpurun_out/r3l_profile.log
