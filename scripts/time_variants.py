"""times the exact walk launch of cfg 2 for each library in build_variants/ matching a prefix
(run on the GPU box): python scripts/time_variants.py libn2v_rev [p q]"""
import glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", ".")
prefix = sys.argv[1]
pq = sys.argv[2:4] if len(sys.argv) >= 4 else ["0.5", "2.0"]
code = f"""
import os, sys, time, torch
sys.path.insert(0, {root!r})
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)[:47104].contiguous()
best = 1e9
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    walks, valid = rw.walk(g, start, 10, 80, {pq[0]}, {pq[1]}, 42)
    torch.cuda.synchronize(); best = min(best, time.time() - t)
print(f"{{best*1e3:8.1f}} ms")
"""
for lib in [None] + sorted(glob.glob(os.path.join(root, "build_variants", prefix + "*.so"))):
    env = dict(os.environ)
    if lib:
        env["N2V_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) if lib else "in-tree", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
