#!/bin/bash
# counters of the batched (and default) SGNS kernel on the cfg3 corpus: separate --pmc passes
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3_pmc_batched}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/r3/time_batched.py cfg3 128"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_0trace -- $CMD > $OUT/p0.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_1 -- $CMD > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_2 -- $CMD > $OUT/p2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_3 -- $CMD > $OUT/p3.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_4 -- $CMD > $OUT/p4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_5 -- $CMD > $OUT/p5.log 2>&1 || exit 1
python3 $R/scripts/condense_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +4M -delete
find $OUT -name "*.db" -delete
tail -5 $OUT/p0.log
