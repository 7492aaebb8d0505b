# GPU call 30: non-dyadic p, q through the all-tables kernel (instance 2): parity, fuzz, timings
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_wedge_gpu.py tests/test_api_gpu.py -x -q > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 150 1201 > $O/fuzz_walk.log 2>&1
tail -1 $O/fuzz_walk.log; grep -q "fuzz ok" $O/fuzz_walk.log || exit 1
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 120 1202 > $O/fuzz_walk_extreme.log 2>&1
tail -1 $O/fuzz_walk_extreme.log; grep -q "fuzz ok" $O/fuzz_walk_extreme.log || exit 1
GRAPH=cfg4 PQ="3.0,0.7;0.7,3.0;1.3,1.3;3.0,1.0;0.7,0.3" timeout -k 10 500 python scripts/time_wedge_kernel.py "r03e" > $O/time.log 2>&1 || exit 1
grep exact $O/time.log
