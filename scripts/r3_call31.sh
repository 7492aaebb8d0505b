# GPU call 31: generic kernel (weighted graphs): clear accepts without the serial row sum
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03f
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_api_gpu.py tests/test_transformers_gpu.py -x -q > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 150 1301 > $O/fuzz_walk.log 2>&1
tail -1 $O/fuzz_walk.log; grep -q "fuzz ok" $O/fuzz_walk.log || exit 1
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 100 1302 > $O/fuzz_walk_extreme.log 2>&1
tail -1 $O/fuzz_walk_extreme.log; grep -q "fuzz ok" $O/fuzz_walk_extreme.log || exit 1
timeout -k 10 300 python scripts/time_weighted.py > $O/time_weighted.log 2>&1 || exit 1
cat $O/time_weighted.log
