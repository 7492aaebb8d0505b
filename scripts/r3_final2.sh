# after splitting the alone-overfull regimes into instance 3: whole GPU suite, smoke, fuzz, timings
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03zz
mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $O/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1
rc=$?; tail -3 $O/tests_gpu.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 1
tail -1 $O/smoke.log
timeout -k 10 300 python scripts/fuzz_walk.py 100 1601 > $O/fuzz_walk.log 2>&1
tail -1 $O/fuzz_walk.log; grep -q "fuzz ok" $O/fuzz_walk.log || exit 1
GRAPH=cfg4 PQ="0.5,2.0;4.0,0.25;0.25,0.25;2.0,2.0" timeout -k 10 400 python scripts/time_wedge_kernel.py "r03zz" > $O/time.log 2>&1 || exit 1
grep exact $O/time.log
timeout -k 10 400 python bench.py --no-sgns --cpu-seconds 3 > $O/bench_walks.json 2> $O/bench.err || exit 1
python3 -c "
import json
d = json.load(open('$O/bench_walks.json'))
print('value %.4g ms %.2f' % (d['value'], d['ms_per_step']), 'biased %.4g ms %.2f' % (d['biased']['value'], d['biased']['ms_per_step']), 'fast %.4g' % d['fast_mode']['value'])
"
