#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --config cfg3 --no-cpu-baseline > gpurun_out/r3u_bench_cfg3_torchrun.json 2> gpurun_out/r3u_bench_cfg3_torchrun.err
echo "torchrun bench rc=$?"; tail -c 300 gpurun_out/r3u_bench_cfg3_torchrun.json
timeout -k 10 300 python bench.py --config cfg2 --steps 10 --warmup 2 > gpurun_out/r3u_bench_cfg2.json 2> gpurun_out/r3u_bench_cfg2.err
echo "cfg2 bench rc=$?"; tail -c 200 gpurun_out/r3u_bench_cfg2.json
