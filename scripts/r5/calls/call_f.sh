# round 5, call f: weighted lanes (tests + timing), big-row pairing stats
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_weighted_lanes_gpu.py tests/test_walk_gpu.py -x -q > gpurun_out/r7f_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/r7f_tests_wlanes.log; exit 1; }
tail -3 gpurun_out/r7f_tests_wlanes.log
KINDS=fp32,fp64 PQ="0.5,2.0;3.0,0.7" timeout -k 10 400 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s\|tables" | tee gpurun_out/r7f_time_wlanes.log
for V in big20k big65k; do
  N2V_VARIANT_LIB=$PWD/build_variants/libn2v_wedge_$V.so timeout -k 10 300 python scripts/r5/big_stats.py 2>&1 | grep "pairings" | tee -a gpurun_out/r7f_big_stats.log
done
