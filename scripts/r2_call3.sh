# GPU call 3 of round 2: whole GPU suite with the hop table, bench on cfg 4, phase statistics
# of the class-count kernel on cfg 4 (diagnostic -DN2V_STATS build)
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/tests_gpu.log 2>&1
rc=$?
echo "tests_exit=$rc" >> $OUT/tests_gpu.log
tail -25 $OUT/tests_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --cpu-seconds 8 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err || exit 1
python3 -c "
import json
d = json.load(open('$OUT/bench_cfg4.json'))
print('value', d['value'], d['roofline']['kernel_ms'], d['roofline']['hop_table'])
for k in ('biased', 'fast_mode', 'sgns'):
    print(k, d[k]['value'], d[k]['ms_per_step'])
print(d['setup'])
"
bash scripts/build_stats.sh > $OUT/build_stats.log 2>&1 || exit 1
GRAPH=cfg4 PQ=0.5,2.0 KERNEL=lanes timeout -k 10 300 python scripts/walk_stats.py > $OUT/walk_stats_cfg4_lanes.log 2>&1
cat $OUT/walk_stats_cfg4_lanes.log
