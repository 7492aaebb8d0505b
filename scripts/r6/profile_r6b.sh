# Runs on the GPU box (gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py with the legs on the graph
# trimmed at the reference's cap of 100 000 only (the exact slots kernel on folded slots).
# usage: bash scripts/r6/profile_r6b.sh <tag>     (outputs under gpurun_out/<tag>/)
set -x
TAG=${1:-r13cap}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-regimes --no-hub --no-audition --no-api --no-weighted --no-sgns --no-fast --no-biased $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py $ARGS > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
tail -2 $OUT/*.log
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -14 $f; done
python3 $R/scripts/condense_pmc.py $OUT
find $OUT -name "*.csv" -size +4M -delete
du -sh $OUT
