#!/bin/bash
# round 3, call 38: the tree as committed -- whole GPU suite (as the driver runs it), smoke, default bench
set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 60; echo "... $(date +%T)"; done ) &
HB=$!
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3am_tests_gpu.log 2>&1
rc=$?; echo "suite rc=$rc"; tail -3 gpurun_out/r3am_tests_gpu.log
[ $rc -eq 0 ] || { kill $HB; exit 1; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3am_smoke.log 2>&1 || { tail -5 gpurun_out/r3am_smoke.log; kill $HB; exit 1; }
tail -1 gpurun_out/r3am_smoke.log
T0=$(date +%s); python bench.py > gpurun_out/r3am_bench_cfg4.json 2> gpurun_out/r3am_bench_cfg4.err
echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
kill $HB
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3am_bench_cfg4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "frac", d["roofline"]["frac"], "gather frac", d["roofline"]["gather_ceiling"]["frac"])
for k in ("biased", "fast_mode", "sgns"):
    print("  ", k, d[k]["value"], d[k].get("trials_per_step"), d[k].get("sampler", ""))
print("   batched", d["sgns"]["batched"].get("value"), "cpu", d["cpu_baseline"]["value"])
PY
