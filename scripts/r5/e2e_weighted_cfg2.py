"""weighted cfg 2 end to end: fit_streaming with exact biased walks at (0.5, 2) on the R-MAT graph with fp32 weights
U[0.1, 2] -- the walks now run on the step-synchronous kernels (csrc/n2v_walk_wlanes.hip); one epoch, dim 128."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import synthetic, randomwalk as rw
from node2vec_amd.pipeline import fit_streaming
g = synthetic.rmat(20, 5_000_000, device="cuda", weights="uniform")
for lanes_from in (1 << 15, 1 << 62):  # the default; never (the one-launch wave-per-walker kernel, round 4's path)
    rw.WEIGHTED_LANES_MIN_WALKERS = lanes_from
    t = {}
    torch.cuda.synchronize(); t0 = time.time()
    out = fit_streaming(g, {"num_walks": 10, "walk_length": 80, "return_param": 0.5, "inout_param": 2.0},
                        {"min_count": 0, "iter": 1, "size": 128, "negative": 5, "sample": 0.0, "window": 5},
                        random_seed=42, batch_vertices=int(os.environ.get("BATCH_VERTICES", 262144)), timings=t)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"weighted cfg2 end to end, exact walks at (0.5, 2), {'step-synchronous kernels' if lanes_from < (1 << 40) else 'wave-per-walker kernel'}: "
          f"{dt:.2f} s (walks {t.get('walk_s', float('nan')):.2f} s, training {t.get('train_s', float('nan')):.2f} s), "
          f"pairs {out.pairs_trained / 1e9:.2f} G, peak HBM {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
