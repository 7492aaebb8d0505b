set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-sgnsmix}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-fast"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > $OUT/p3.log 2>&1
find $OUT -name "*.csv" -size +8M -delete
