set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
timeout -k 10 300 python -m pytest tests/test_walk_gpu.py tests/test_wedge_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/r4k_tests.log 2>&1 || { tail -30 gpurun_out/r4k_tests.log; exit 1; }
tail -2 gpurun_out/r4k_tests.log
timeout -k 10 200 python scripts/fuzz_walk.py 60 4005 > gpurun_out/r4k_fuzz.log 2>&1 || { tail -20 gpurun_out/r4k_fuzz.log; exit 1; }
tail -1 gpurun_out/r4k_fuzz.log
GRAPH=cfg4 PQ="0.5,2;4,0.25;4,2;2,1" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py lds5 2>&1 | grep -v amdgpu | tee gpurun_out/r4k_time_cfg4.log
for v in nolds6 nolds5; do
  N2V_VARIANT_LIB=$R/build_variants/libn2v_wedge_$v.so GRAPH=cfg4 PQ="0.5,2;4,0.25" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py $v 2>&1 | grep -v amdgpu | tee -a gpurun_out/r4k_time_cfg4.log
done
for c in cfg2 cfg3; do GRAPH=$c timeout -k 10 200 python scripts/r4/visited_degree.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r4k_visited_degree.log; done
WALKS=10 timeout -k 10 400 python scripts/r3/time_partitioned.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4k_time_partitioned_w10.log
