# the whole -m gpu suite as the driver runs it (+ durations), smoke, then the default bench
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r4m}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${TAG}_tests_gpu.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_gpu.log; exit 1; }
tail -18 gpurun_out/${TAG}_tests_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_cfg4.json 2> gpurun_out/${TAG}_bench_cfg4.err || { tail -20 gpurun_out/${TAG}_bench_cfg4.err; exit 1; }
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${TAG}_bench_cfg4.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
b = d.get("biased", {})
print("biased", b.get("value"), b.get("roofline", {}).get("kernel"), b.get("roofline", {}).get("frac"))
print("regimes", {k: v.get("value") for k, v in d.get("biased_other_regimes", {}).items()} if isinstance(d.get("biased_other_regimes"), dict) else d.get("biased_other_regimes"))
s = d.get("sgns", {})
print("sgns", s.get("value"), s.get("hub_rows_auto"), s.get("plain_stores", {}).get("value"), s.get("batched", {}).get("value"))
print("fast", d.get("fast_mode", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
