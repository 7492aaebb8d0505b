# instruction mix / stall counters of the walk kernel (separate PMC passes, no tracing)
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-mix}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -oE "SQ_[A-Z_0-9]+" $OUT/counters.txt | sort -u | tr '\n' ' ' | head -c 3000 > $OUT/sq_names.txt
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-sgns --no-fast"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > $OUT/p3.log 2>&1
tail -2 $OUT/p1.log $OUT/p2.log $OUT/p3.log
find $OUT -name "*.csv" -size +8M -delete
