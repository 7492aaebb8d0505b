# Runs on the GPU box (gpurun): bench + rocprofv3 kernel trace + separate PMC passes.
# usage: bash scripts/profile_walk.sh <tag> [extra bench args]
set -x
TAG=${1:-r01}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -12 $f; done
# keep only what fits the 64 MiB merge budget
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
