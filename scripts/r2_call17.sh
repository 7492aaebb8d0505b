set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02q
(cd /tmp && export TMPDIR=/tmp && GRAPH=cfg5 PQ=4.0,0.25 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/r02q/pmc_sq -- python3 $R/scripts/time_wedge_kernel.py prof > $R/gpurun_out/r02q/pmc.log 2>&1)
python3 scripts/condense_pmc.py gpurun_out/r02q > /dev/null
python3 - <<PY
import json
d=json.load(open("gpurun_out/r02q/pmc_summary.json"))
for k,v in d.get("pmc_sq",{}).items():
    if "wedge_kernel" in k: print(k[-40:], "median %.4g"%v["median"])
PY
find gpurun_out/r02q -name "*.csv" -size +2M -delete
