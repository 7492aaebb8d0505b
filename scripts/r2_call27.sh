# GPU call 21: closed forms in exact fp64 arithmetic, class flags without divisions: parity + timings
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r03a
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_scale_props_gpu.py tests/test_scale_cfg345_gpu.py -x -q > gpurun_out/r03a/tests.log 2>&1
rc=$?; tail -3 gpurun_out/r03a/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 200 777 > gpurun_out/r03a/fuzz_walk.log 2>&1
tail -1 gpurun_out/r03a/fuzz_walk.log; grep -q "fuzz ok" gpurun_out/r03a/fuzz_walk.log || exit 1
FUZZ_PQ=extreme timeout -k 10 200 python scripts/fuzz_walk.py 150 778 > gpurun_out/r03a/fuzz_walk_extreme.log 2>&1
tail -1 gpurun_out/r03a/fuzz_walk_extreme.log; grep -q "fuzz ok" gpurun_out/r03a/fuzz_walk_extreme.log || exit 1
for pq in 0.5,2.0 2.0,2.0 0.25,0.25 2.0,1.0 4.0,2.0; do GRAPH=cfg4 PQ=$pq python scripts/time_wedge_kernel.py "cfg4 $pq"; done > gpurun_out/r03a/time.log 2>&1
for pq in 0.5,2.0 4.0,0.25; do GRAPH=cfg5 PQ=$pq python scripts/time_wedge_kernel.py "cfg5 $pq"; done >> gpurun_out/r03a/time.log 2>&1
GRAPH=cfg2 PQ=0.5,2.0 python scripts/time_wedge_kernel.py "cfg2 0.5,2.0" >> gpurun_out/r03a/time.log 2>&1
grep exact gpurun_out/r03a/time.log
timeout -k 10 600 python bench.py --cpu-seconds 6 --no-sgns > gpurun_out/r03a/bench_cfg4.json 2> gpurun_out/r03a/bench_cfg4.err || exit 1
python3 -c "
import json
d = json.load(open('gpurun_out/r03a/bench_cfg4.json'))
print('value %.4g' % d['value'], 'biased %.4g ms %.2f' % (d['biased']['value'], d['biased']['ms_per_step']), 'fast %.4g' % d['fast_mode']['value'])
"
