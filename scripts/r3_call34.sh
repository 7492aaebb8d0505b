# GPU call 34: mirror closed forms find the next "other" slot from the rank of pick in the list
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03i
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_wedge_gpu.py -x -q > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 120 1401 > $O/fuzz_walk.log 2>&1
tail -1 $O/fuzz_walk.log; grep -q "fuzz ok" $O/fuzz_walk.log || exit 1
FUZZ_PQ=two timeout -k 10 300 python scripts/fuzz_walk.py 100 1402 > $O/fuzz_walk_two.log 2>&1
tail -1 $O/fuzz_walk_two.log; grep -q "fuzz ok" $O/fuzz_walk_two.log || exit 1
GRAPH=cfg4 PQ="4.0,0.25;2.0,0.5;0.5,0.25;0.25,0.25;0.25,0.5;0.5,2.0" timeout -k 10 500 python scripts/time_wedge_kernel.py "r03i" > $O/time.log 2>&1 || exit 1
grep exact $O/time.log
