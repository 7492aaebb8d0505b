#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python scripts/r3/e2e_cfg4.py batched 1.0 > gpurun_out/r3z_e2e_cfg4_batched.json 2> gpurun_out/r3z_e2e_b.err; echo "rc=$?"; cat gpurun_out/r3z_e2e_cfg4_batched.json
timeout -k 10 750 python scripts/r3/e2e_cfg4.py default 1.0 > gpurun_out/r3z_e2e_cfg4_default.json 2> gpurun_out/r3z_e2e_d.err; echo "rc=$?"; cat gpurun_out/r3z_e2e_cfg4_default.json
