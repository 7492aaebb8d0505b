import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic
from node2vec_amd.pipeline import fit_streaming
g = synthetic.rmat(20, 5_000_000, device="cuda")
for mode in ("exact", "fast"):
    torch.cuda.synchronize(); t = time.time()
    out = fit_streaming(g, {"num_walks": 10, "walk_length": 80, "return_param": 0.5, "inout_param": 2.0},
                        {"min_count": 0, "iter": 1, "size": 128, "negative": 5, "sample": 0.0, "window": 5},
                        random_seed=42, batch_vertices=65536, mode=mode)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"cfg2 end to end ({mode} walks, 1 epoch, dim 128): {dt:.2f} s, vocabulary {len(out.wv.vocab)}, pairs {out.pairs_trained/1e9:.2f} G, "
          f"peak HBM {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
