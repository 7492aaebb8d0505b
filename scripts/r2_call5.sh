# GPU call 5: new parity tests (tolerances to be calibrated), partitioned walking on the GPU,
# full GPU suite, walk fuzz against the oracle with the hop table, bench
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_sgns_parity_gpu.py tests/test_partitioned_gpu.py -x -q -s --durations=5 > $OUT/tests_parity.log 2>&1
echo "tests_exit=$?" >> $OUT/tests_parity.log
grep -E "karate:|rmat-1m:|passed|failed|Error|assert" $OUT/tests_parity.log | head -30
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_sgns_parity_gpu.py > $OUT/tests_gpu.log 2>&1
echo "tests_exit=$?" >> $OUT/tests_gpu.log
tail -15 $OUT/tests_gpu.log
timeout -k 10 400 python scripts/fuzz_walk.py 240 777 > $OUT/fuzz_walk.log 2>&1
tail -3 $OUT/fuzz_walk.log
timeout -k 10 200 python scripts/fuzz_sgns.py 90 778 > $OUT/fuzz_sgns.log 2>&1
tail -2 $OUT/fuzz_sgns.log
