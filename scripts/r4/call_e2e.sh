set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 400 python scripts/r3/e2e_cfg4.py batched > gpurun_out/r5e_e2e_cfg4_batched.json 2> gpurun_out/r5e_e2e_batched.err || { tail -5 gpurun_out/r5e_e2e_batched.err; exit 1; }
cat gpurun_out/r5e_e2e_cfg4_batched.json
timeout -k 10 750 python scripts/r3/e2e_cfg4.py default > gpurun_out/r5e_e2e_cfg4_default.json 2> gpurun_out/r5e_e2e_default.err || { tail -5 gpurun_out/r5e_e2e_default.err; exit 1; }
cat gpurun_out/r5e_e2e_cfg4_default.json
