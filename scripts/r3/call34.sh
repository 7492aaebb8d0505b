#!/bin/bash
# round 3, call 34: the last build -- whole GPU suite as the driver runs it, smoke, default bench,
# cfg 3 bench, then the rocprofv3 profile (kernel trace + PMC passes)
set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 60; echo "... $(date +%T)"; done ) &
HB=$!
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3ah_tests_gpu.log 2>&1
rc=$?; echo "suite rc=$rc"; tail -3 gpurun_out/r3ah_tests_gpu.log
[ $rc -eq 0 ] || { kill $HB; exit 1; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3ah_smoke.log 2>&1 || { tail -5 gpurun_out/r3ah_smoke.log; kill $HB; exit 1; }
tail -1 gpurun_out/r3ah_smoke.log
T0=$(date +%s); python bench.py > gpurun_out/r3ah_bench_cfg4.json 2> gpurun_out/r3ah_bench_cfg4.err
echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r3ah_bench_cfg3.json 2> gpurun_out/r3ah_bench_cfg3.err
echo "bench cfg3 rc=$?"
timeout -k 10 900 bash scripts/r3/profile_r3.sh r3ah_cfg4 > gpurun_out/r3ah_profile.log 2>&1
echo "profile rc=$?"
kill $HB
python3 - <<'PY'
import json
for f in ("gpurun_out/r3ah_bench_cfg4.json", "gpurun_out/r3ah_bench_cfg3.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, {k: d[k] for k in ("value", "ms_per_step")}, "frac", d["roofline"]["frac"], "gather frac", d["roofline"]["gather_ceiling"]["frac"])
    for k in ("biased", "fast_mode", "sgns"):
        if k in d: print("  ", k, d[k]["value"], d[k].get("trials_per_step"))
    if "sgns" in d and "batched" in d["sgns"]: print("   batched", d["sgns"]["batched"].get("value"))
PY
