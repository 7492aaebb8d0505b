#!/bin/bash
# round 3, call 35: the differential fuzzers on the last build
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python scripts/fuzz_walk.py 240 21 > gpurun_out/r3aj_fuzz_walk.log 2>&1; echo "walk rc=$?"; tail -1 gpurun_out/r3aj_fuzz_walk.log
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 120 22 > gpurun_out/r3aj_fuzz_walk_extreme.log 2>&1; echo "extreme rc=$?"; tail -1 gpurun_out/r3aj_fuzz_walk_extreme.log
FUZZ_PQ=two FUZZ_PARTITIONED=1 timeout -k 10 300 python scripts/fuzz_walk.py 120 23 > gpurun_out/r3aj_fuzz_walk_two_partitioned.log 2>&1; echo "two rc=$?"; tail -1 gpurun_out/r3aj_fuzz_walk_two_partitioned.log
timeout -k 10 400 python scripts/fuzz_sgns.py 200 24 > gpurun_out/r3aj_fuzz_sgns.log 2>&1; echo "sgns rc=$?"; tail -1 gpurun_out/r3aj_fuzz_sgns.log
