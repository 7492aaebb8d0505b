# GPU call 15: why is the mirror case slow on cfg 5?  (p, q) variants on the cfg 5 graph; SGNS dim 256 at 50 M rows
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02o
for pq in 0.5,2.0 4.0,0.25 2.0,0.5; do GRAPH=cfg5 PQ=$pq python scripts/time_wedge_kernel.py "cfg5 $pq"; done > gpurun_out/r02o/time_cfg5.log 2>&1
grep exact gpurun_out/r02o/time_cfg5.log
(cd /tmp && export TMPDIR=/tmp && GRAPH=cfg5 PQ=4.0,0.25 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02o/trace -- python3 $R/scripts/time_wedge_kernel.py prof > $R/gpurun_out/r02o/trace.log 2>&1)
for f in $(find gpurun_out/r02o/trace -name "*kernel_stats.csv"); do head -6 $f | cut -c1-160; done
python scripts/time_sgns_scale.py 5e7 256 > gpurun_out/r02o/sgns_50m_256.log 2>&1; grep Mpairs gpurun_out/r02o/sgns_50m_256.log
python scripts/time_sgns_scale.py 1e8 128 >> gpurun_out/r02o/sgns_50m_256.log 2>&1; grep Mpairs gpurun_out/r02o/sgns_50m_256.log
find gpurun_out/r02o -name "*.csv" -size +2M -delete
