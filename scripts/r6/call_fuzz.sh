# round 6: the differential fuzzers on the final tree (all modes, ~18 min): unit / weighted / partitioned walks and
# SGNS against the oracle, the weighted margin kernels against the wave-per-walker kernel
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r10u}
O=gpurun_out/${TAG}_fuzz.log
: > $O
FUZZ_PARTITIONED=1 timeout -k 10 400 python scripts/fuzz_walk.py 240 91 2>&1 | tail -3 >> $O; echo "-- partitioned done" >> $O
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 150 92 2>&1 | tail -3 >> $O
FUZZ_PQ=rational timeout -k 10 300 python scripts/fuzz_walk.py 150 93 2>&1 | tail -3 >> $O
FUZZ_PQ=two timeout -k 10 300 python scripts/fuzz_walk.py 120 94 2>&1 | tail -3 >> $O
timeout -k 10 300 python scripts/fuzz_walk.py 150 95 2>&1 | tail -3 >> $O
timeout -k 10 200 python scripts/fuzz_sgns.py 90 96 2>&1 | tail -2 >> $O
timeout -k 10 400 python scripts/r5/fuzz_weighted_margins.py 240 97 2>&1 | tail -3 >> $O
cat $O
