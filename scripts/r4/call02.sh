# where does the time of the passes go?  kernel trace of one (p, q) + occupancy variants
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out
for v in w5 w6; do
  N2V_VARIANT_LIB=$R/build_variants/libn2v_w2_$v.so GRAPH=cfg4 PQ="0.5,2" ROUNDS="4" timeout -k 10 300 python scripts/r4/time_wedge2.py $v 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c_time_variants.log
done
cd /tmp && export TMPDIR=/tmp
GRAPH=cfg4 PQ="0.5,2" ROUNDS="4" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4c_trace -- python3 $R/scripts/r4/time_wedge2.py w7 > $R/gpurun_out/r4c_trace.log 2>&1
for f in $(find $R/gpurun_out/r4c_trace -name "*kernel_stats.csv"); do head -25 $f; cp $f $R/gpurun_out/r4c_kernel_stats.csv; done
find $R/gpurun_out/r4c_trace -name "*.csv" -size +4M -delete
