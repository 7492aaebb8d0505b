# end-of-round verification as the driver does it: build check, full GPU suite, smoke, default bench
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r03v
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > gpurun_out/r03v/build.log 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r03v/tests_gpu.log 2>&1
rc=$?; tail -12 gpurun_out/r03v/tests_gpu.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03v/smoke.log 2>&1 || exit 1
tail -1 gpurun_out/r03v/smoke.log
timeout -k 10 700 python bench.py > gpurun_out/r03v/bench.json 2> gpurun_out/r03v/bench.err || exit 1
python3 -c "
import json
d = json.load(open('gpurun_out/r03v/bench.json'))
print('value %.4g frac %.3f traffic %s' % (d['value'], d['roofline']['frac'], d['roofline']['traffic']))
for k in ('biased', 'fast_mode'):
    print(k, '%.4g' % d[k]['value'], 'frac %.3f' % d[k]['roofline']['frac'], d[k]['roofline']['traffic'], d[k]['roofline'].get('gather_ceiling', {}).get('frac'))
print('sgns %.4g frac %.3f' % (d['sgns']['value'], d['sgns']['roofline']['frac']))
"
