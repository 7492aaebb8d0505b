# round 6 (second session): the passes over a workspace (make WEDGE2=1: closed forms in the main launches, declined
# steps replayed 64 to a wave) on cfg 4 trimmed at 100 000, where a replay costs a wave half a millisecond
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r12d_wedge2_cap100000.log
for rounds in 2 4; do
N2V_WEDGE2_ROUNDS=$rounds USE_WS=1 N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge2.so PQ="4,0.25;0.5,2" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py passes_rounds$rounds >> gpurun_out/r12d_wedge2_cap100000.log 2>&1 || { tail -30 gpurun_out/r12d_wedge2_cap100000.log; exit 1; }
done
N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge2.so PQ="4,0.25;0.5,2" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py one_launch >> gpurun_out/r12d_wedge2_cap100000.log 2>&1
grep "G steps" gpurun_out/r12d_wedge2_cap100000.log
