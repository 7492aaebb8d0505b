# GPU call 11: the all-tables kernel (n2v_walk_wedge.hip): tests, fuzz, occupancy variants
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02k
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
timeout -k 10 500 bash scripts/wedge_waves.sh > $OUT/wedge_waves.log 2>&1
grep -v amdgpu.ids $OUT/wedge_waves.log | tail -6
