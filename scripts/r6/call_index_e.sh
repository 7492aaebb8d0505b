# round 6 (second session), call e: the slots kernel with its slow steps set aside -- tests, then cfg 4 at both caps
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_index_gpu.py tests/test_wedge_gpu.py tests/test_walk_gpu.py -x -q > gpurun_out/r11e_tests.log 2>&1 || { tail -40 gpurun_out/r11e_tests.log; exit 1; }
tail -3 gpurun_out/r11e_tests.log
PARK=1 PQ="0.5,2;4,0.25;3,0.7" timeout -k 10 400 python scripts/r6/time_wedge_index.py r11e > gpurun_out/r11e_time_park_cap100000.log 2>&1 || { tail -30 gpurun_out/r11e_time_park_cap100000.log; exit 1; }
grep "G \|wedge_mode" gpurun_out/r11e_time_park_cap100000.log
PARK=1 TRIM=10000 PQ="0.5,2;4,0.25;3,0.7" timeout -k 10 400 python scripts/r6/time_wedge_index.py r11e > gpurun_out/r11e_time_park_cap10000.log 2>&1 || { tail -30 gpurun_out/r11e_time_park_cap10000.log; exit 1; }
grep "G \|wedge_mode" gpurun_out/r11e_time_park_cap10000.log
