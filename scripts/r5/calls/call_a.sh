# round 5, call a: the self-launch tests + the default bench with the reordered line
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py -x -q -k "bench" > gpurun_out/r7a_tests_bench.log 2>&1 || { tail -40 gpurun_out/r7a_tests_bench.log; exit 1; }
tail -3 gpurun_out/r7a_tests_bench.log
timeout -k 10 900 python bench.py > gpurun_out/r7a_bench_cfg4.json 2> gpurun_out/r7a_bench_cfg4.err || { tail -20 gpurun_out/r7a_bench_cfg4.err; exit 1; }
python - <<PY
import json
line = [l for l in open("gpurun_out/r7a_bench_cfg4.json") if l.startswith("{")][-1]
d = json.loads(line)
print(len(line), list(d))
print(json.dumps(d["summary"], indent=1))
print(d["setup"])
print(len(json.dumps({k: d[k] for k in list(d)[list(d).index("biased"):]})))
PY
