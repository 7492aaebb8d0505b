#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python scripts/fuzz_sgns.py 240 31 > gpurun_out/r3r_fuzz_sgns.log 2>&1; tail -2 gpurun_out/r3r_fuzz_sgns.log
timeout -k 10 300 python scripts/fuzz_walk.py 150 31 > gpurun_out/r3r_fuzz_walk.log 2>&1; tail -2 gpurun_out/r3r_fuzz_walk.log
