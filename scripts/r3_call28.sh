# GPU call 28: return slot sharing a stack with "other" (p > q > 1, p < q < 1): parity, fuzz, timings;
# (the variant library was the build with the rare replays inline, then switchable by -DN2V_REPLAY_INLINE)
# replays out of line against inline
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03b
mkdir -p $O build_stats
(cd node2vec_amd/csrc && /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off -shared -o ../../build_stats/libn2v_inl.so n2v_capi.hip n2v_walk.hip n2v_walk_unit.hip n2v_walk_fast.hip n2v_walk_uniform.hip n2v_alias.hip n2v_sgns.hip n2v_trim.hip n2v_edge_classes.hip n2v_sync.hip n2v_transform.hip n2v_hops.hip n2v_wedge.hip n2v_walk_wedge.hip) > $O/variant_build.log 2>&1 &
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_wedge_gpu.py -x -q > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
FUZZ_PQ=two timeout -k 10 300 python scripts/fuzz_walk.py 180 881 > $O/fuzz_walk_two.log 2>&1
tail -1 $O/fuzz_walk_two.log; grep -q "fuzz ok" $O/fuzz_walk_two.log || exit 1
timeout -k 10 300 python scripts/fuzz_walk.py 150 882 > $O/fuzz_walk.log 2>&1
tail -1 $O/fuzz_walk.log; grep -q "fuzz ok" $O/fuzz_walk.log || exit 1
wait
GRAPH=cfg4 PQ="0.5,2.0;4.0,2.0;0.25,0.5;4.0,0.25;2.0,2.0" timeout -k 10 400 python scripts/time_wedge_kernel.py "out-of-line" > $O/time.log 2>&1 || exit 1
grep exact $O/time.log
N2V_VARIANT_LIB=$R/build_stats/libn2v_inl.so GRAPH=cfg4 PQ="0.5,2.0;4.0,2.0;0.25,0.5;4.0,0.25;2.0,2.0" timeout -k 10 400 python scripts/time_wedge_kernel.py "inline" >> $O/time.log 2>&1 || exit 1
grep exact $O/time.log
