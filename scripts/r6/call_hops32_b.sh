# round 6 (second session): the half slot behind a hop entry loaded only where the picked edge has a list
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
N2V_HOPS32=1 timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q > gpurun_out/r14f_tests.log 2>&1 || { tail -40 gpurun_out/r14f_tests.log; exit 1; }
tail -1 gpurun_out/r14f_tests.log
N2V_HOPS32=1 timeout -k 10 200 python scripts/fuzz_walk.py 60 23 > gpurun_out/r14f_fuzz.log 2>&1 || { tail -30 gpurun_out/r14f_fuzz.log; exit 1; }
tail -1 gpurun_out/r14f_fuzz.log
: > gpurun_out/r14f_time.log
for h in 1 0; do N2V_HOPS32=$h timeout -k 10 400 python bench.py --config cfg5 --no-sgns --no-api --no-weighted --no-fast --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg5 hops32=$h', {k:round(v/1e9,2) for k,v in d['summary']['walk_steps_per_s'].items() if v})" >> gpurun_out/r14f_time.log; done
for h in 0 1 0 1; do
  N2V_HOPS32=$h TRIM=10000 PQ="0.5,2;4,0.25;3,0.7;0.25,0.5" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py hops32_$h 2>&1 | grep "G steps" >> gpurun_out/r14f_time.log
done
cat gpurun_out/r14f_time.log
