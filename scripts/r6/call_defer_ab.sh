# A/B: product (deferred hop gather on inline-return-position edges) vs -DN2V_DEFER_HOP=0, exact biased legs of bench.py
R=$GRAFT_REPO_ROOT
CFG=${1:-cfg4}
ARGS="--config $CFG --steps 6 --warmup 2 --no-sgns --no-fast --no-weighted --no-api --no-ref-cap --no-cpu-baseline"
for lib in "" $R/build_variants/libn2v_defer${DEFER:-0}.so; do
  N2V_HIP_LIB=$lib timeout -k 10 500 python $R/bench.py $ARGS > $R/gpurun_out/r10o_ab_$CFG$( [ -n "$lib" ] && echo _variant ).json 2>/dev/null || exit 1
done
python3 - <<PY
import json
for tag in ("", "_variant"):
    d = json.load(open("$R/gpurun_out/r10o_ab_$CFG%s.json" % tag))
    w = d["summary"]["walk_steps_per_s"]
    print("$CFG", tag or "defer  ", {k: round(v / 1e9, 2) for k, v in w.items() if v and k.startswith(("exact", "headline"))})
PY
