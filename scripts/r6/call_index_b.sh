# round 6 (second session), call b: cfg 4 trimmed at the reference's cap -- the searches cut by parallel probes
# (variants of n2v_walk_wedge.hip), with and without the index; what the steps of (3, 0.7) do
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python scripts/r6/near_count_cap.py > gpurun_out/r11b_near_count_cap100000.log 2>&1 || { tail -30 gpurun_out/r11b_near_count_cap100000.log; exit 1; }
cat gpurun_out/r11b_near_count_cap100000.log
for v in base k8 k16 tail; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib PQ="0.5,2;4,0.25" timeout -k 10 300 python scripts/r6/time_wedge_index.py r11b_$v > gpurun_out/r11b_time_$v.log 2>&1 || { tail -30 gpurun_out/r11b_time_$v.log; exit 1; }
  grep "G steps\|wedge_mode" gpurun_out/r11b_time_$v.log
done
