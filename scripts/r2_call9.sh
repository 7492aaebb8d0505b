# GPU call 6: wedge table -- build test, bit-parity tests, fuzz, stats and bench
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02i
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/tests.log 2>&1
rc=$?
echo "tests_exit=$rc" >> $OUT/tests.log
tail -25 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 240 31337 > $OUT/fuzz_walk.log 2>&1
tail -3 $OUT/fuzz_walk.log
grep -q "fuzz ok" $OUT/fuzz_walk.log || exit 1
bash scripts/build_stats.sh > $OUT/build_stats.log 2>&1 || exit 1
GRAPH=cfg4 PQ=0.5,2.0 KERNEL=lanes timeout -k 10 300 python scripts/walk_stats.py > $OUT/walk_stats_cfg4_lanes.log 2>&1
cat $OUT/walk_stats_cfg4_lanes.log
timeout -k 10 600 python bench.py --cpu-seconds 6 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err || exit 1
python3 -c "
import json
d = json.load(open('$OUT/bench_cfg4.json'))
print('value', d['value'], d['roofline']['kernel_ms'])
for k in ('biased', 'fast_mode', 'sgns'):
    print(k, d[k]['value'], d[k]['ms_per_step'])
print(d['setup'])
"
