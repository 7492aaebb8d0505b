# the default bench (as the driver runs it) and the line's summary
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7l}; shift || true
timeout -k 10 900 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/${TAG}_bench_cfg4.json 2> gpurun_out/${TAG}_bench_cfg4.err || { tail -20 gpurun_out/${TAG}_bench_cfg4.err; exit 1; }
python - <<PY
import json
line = [l for l in open("gpurun_out/${TAG}_bench_cfg4.json") if l.startswith("{")][-1]
d = json.loads(line)
print(len(line), "chars; tail from 'biased':", len(json.dumps({k: d[k] for k in list(d)[list(d).index("biased"):]})) if "biased" in d else None)
print(json.dumps(d["summary"], indent=1))
print("audition", d["roofline"].get("output_buffer_audition"), d.get("roofline_vertex_ids", {}).get("output_buffer_audition"))
print("setup", {k: v for k, v in d["setup"].items() if not isinstance(v, dict)})
PY
