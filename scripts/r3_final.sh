# end of round: the driver's sequence (build, whole -m gpu suite, smoke, default bench) + fuzz of the final build
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03z
mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $O/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/tests_gpu.log 2>&1
rc=$?; tail -12 $O/tests_gpu.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 1
tail -1 $O/smoke.log
timeout -k 10 700 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
python3 -c "
import json
d = json.load(open('$O/bench.json'))
print('value %.4g frac %.3f traffic %s' % (d['value'], d['roofline']['frac'], d['roofline']['traffic']))
for k in ('biased', 'fast_mode'):
    print(k, '%.4g' % d[k]['value'], 'frac %.3f' % d[k]['roofline']['frac'], d[k]['roofline']['traffic'], d[k]['roofline'].get('gather_ceiling', {}).get('frac'))
print('sgns %.4g frac %.3f' % (d['sgns']['value'], d['sgns']['roofline']['frac']))
"
timeout -k 10 300 python scripts/fuzz_walk.py 200 1501 > $O/fuzz_walk.log 2>&1
tail -1 $O/fuzz_walk.log; grep -q "fuzz ok" $O/fuzz_walk.log || exit 1
FUZZ_PQ=two timeout -k 10 300 python scripts/fuzz_walk.py 150 1502 > $O/fuzz_walk_two.log 2>&1
tail -1 $O/fuzz_walk_two.log; grep -q "fuzz ok" $O/fuzz_walk_two.log || exit 1
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 150 1503 > $O/fuzz_walk_extreme.log 2>&1
tail -1 $O/fuzz_walk_extreme.log; grep -q "fuzz ok" $O/fuzz_walk_extreme.log || exit 1
timeout -k 10 300 python scripts/fuzz_sgns.py 100 1504 > $O/fuzz_sgns.log 2>&1
tail -1 $O/fuzz_sgns.log
