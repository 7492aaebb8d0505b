#!/bin/bash
# round 3, call 31: smoke + the default bench line of the final build, timed as the driver does
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3ae_smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/r3ae_smoke.log
T0=$(date +%s); python bench.py > gpurun_out/r3ae_bench_cfg4.json 2> gpurun_out/r3ae_bench_cfg4.err
echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3ae_bench_cfg4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup")})
print("roofline", d["roofline"])
print("cpu_baseline", d["cpu_baseline"])
for k in ("biased", "fast_mode", "sgns"):
    if k in d: print(k, json.dumps(d[k])[:600])
PY
