# GPU call 33: the exchange passes on the device; bench.py as the driver launches it for N > 1
# (torch.distributed.run, RCCL), here with one rank
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03h
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_delta_sync_gpu.py -x -q > $O/tests.log 2>&1
rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench_torchrun.json 2> $O/bench_torchrun.err
rc=$?; tail -3 $O/bench_torchrun.err; [ $rc -eq 0 ] || exit 1
python3 -c "
import json
d = json.loads([l for l in open('$O/bench_torchrun.json') if l.startswith('{')][-1])
print('n_gpus', d['n_gpus'], 'value %.4g' % d['value'], 'sgns %.4g' % d['sgns']['value'], d['sgns'].get('exchange'), 'cpu_baseline' in d)
"
