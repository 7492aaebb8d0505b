set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r4y}
timeout -k 10 600 python -m pytest tests/test_api_gpu.py tests/test_ranked_gpu.py tests/test_multirank_gpu.py -x -q > gpurun_out/${TAG}_tests_api.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_api.log; exit 1; }
tail -3 gpurun_out/${TAG}_tests_api.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_cfg4.json 2> gpurun_out/${TAG}_bench_cfg4.err || { tail -20 gpurun_out/${TAG}_bench_cfg4.err; exit 1; }
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${TAG}_bench_cfg4.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("gather_ceiling"))
v = d.get("vertex_id_output", {})
print("vertex ids", v.get("value"), v.get("roofline", {}).get("frac"))
b = d.get("biased", {})
print("biased", b.get("value"), b.get("roofline", {}).get("kernel"), b.get("roofline", {}).get("frac"))
s = d.get("sgns", {})
print("sgns", s.get("value"), s.get("hub_rows_auto"), s.get("plain_stores", {}).get("value"), s.get("batched", {}).get("value"))
print("fast", d.get("fast_mode", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"))
print(d["setup"])
PY
