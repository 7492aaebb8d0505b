# GPU call 11: the all-tables kernel (n2v_walk_wedge.hip): tests, fuzz, occupancy variants
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02m
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_scale_props_gpu.py tests/test_scale_cfg345_gpu.py tests/test_api_gpu.py -x -q > $OUT/tests.log 2>&1
rc=$?
echo "tests_exit=$rc" >> $OUT/tests.log
tail -6 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 150 555 > $OUT/fuzz_walk.log 2>&1
tail -2 $OUT/fuzz_walk.log
grep -q "fuzz ok" $OUT/fuzz_walk.log || exit 1
FUZZ_PQ=extreme timeout -k 10 200 python scripts/fuzz_walk.py 90 556 > $OUT/fuzz_walk_extreme.log 2>&1
tail -2 $OUT/fuzz_walk_extreme.log
grep -q "fuzz ok" $OUT/fuzz_walk_extreme.log || exit 1
for pq in 4.0,0.25 2.0,0.5 0.25,0.25 0.5,2.0; do PQ=$pq GRAPH=cfg2 python scripts/time_wedge_kernel.py "cfg2 $pq"; done > $OUT/time_pq.log 2>&1; GRAPH=cfg5 PQ=4.0,0.25 python scripts/time_wedge_kernel.py "cfg5 4,0.25" >> $OUT/time_pq.log 2>&1
grep -v amdgpu.ids $OUT/time_pq.log
