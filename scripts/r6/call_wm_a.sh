# round 6 (second session): the block summaries of the weighted wave kernel searched in two round trips instead of nine
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_weighted_lanes_gpu.py tests/test_margin_adversary_gpu.py -x -q > gpurun_out/r13t_tests_weighted.log 2>&1 || { tail -40 gpurun_out/r13t_tests_weighted.log; exit 1; }
tail -1 gpurun_out/r13t_tests_weighted.log
timeout -k 10 300 python scripts/r5/fuzz_weighted_margins.py 120 21 2>&1 | tail -1
for rep in 1 2; do
for v in new kary0; do
  lib=$PWD/node2vec_amd/libn2v_hip.so
  [ $v = kary0 ] && lib=$PWD/build_variants/libn2v_wlanes_kary0.so
  N2V_HIP_LIB=$lib timeout -k 10 400 python bench.py --no-sgns --no-api --no-fast --no-biased --no-ref-cap --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); w=d['weighted']; print('$v', w['value'], {k:v['value'] for k,v in w['small_batches'].items() if isinstance(v,dict) and 'value' in v})"
done
done
