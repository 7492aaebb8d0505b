# GPU call 4 of round 2: profiles (kernel trace + PMC passes) of the hop-table / cum-index build on
# cfg 4 and cfg 3, size of per-edge shared-position lists, new GPU tests
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_indexer_gpu.py tests/test_transformers_gpu.py tests/test_sgns_gpu.py -x -q -k "not statistical" > $OUT/tests_new.log 2>&1
echo "tests_exit=$?" >> $OUT/tests_new.log
tail -8 $OUT/tests_new.log
timeout -k 10 300 python scripts/wedge_size.py > $OUT/wedge_size.log 2>&1
cat $OUT/wedge_size.log
bash scripts/profile_r2.sh r02c_cfg4 --config cfg4 || exit 1
bash scripts/profile_r2.sh r02c_cfg3 --config cfg3 || exit 1
du -sh $R/gpurun_out
