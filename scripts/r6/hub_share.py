"""For the corpus of a BASELINE config: the rows SgnsModel.auto_hub_rows selects (lambda >= 1.5) and the share of all
row-holds (token share + k x negative-draw share, over 1 + k) they carry.   python scripts/r6/hub_share.py cfg3 [cfg4 cfg5 cfg2]"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import bench
from node2vec_amd import randomwalk as rw
from node2vec_amd.pipeline import corpus_vocabulary
for name in sys.argv[1:] or ["cfg3"]:
    cfg = bench.CONFIGS[name]
    g = bench.build_graph(cfg, torch, torch.device("cuda"), {})
    start_all = rw.start_vertices(g)
    p, q = cfg["p"], cfg["q"]
    in_ranks = p == 1.0 and q == 1.0
    if in_ranks:
        g.build_ranked()
    B = 1 << 20
    def count_walk(k):
        return rw.walk(g, start_all[k * B:(k + 1) * B], 10, 80, p, q, 42, mode="exact", check=False, rank_ids=in_ranks)
    vocab, _ = corpus_vocabulary(g, count_walk, -(-start_all.numel() // B), 0, in_ranks)
    c = vocab.counts.double()
    f = c / c.sum(); pw = c.pow(0.75); n = pw / pw.sum()
    for waves in (8192, 4096):
        lam = waves * (f + 5 * n)
        held = (f + 5 * n) / 6
        H = int((lam >= 1.5).sum())
        print(f"{name}: vocab {c.numel()}, waves {waves}: H(lambda >= 1.5) = {H}, share of all row-holds {float(held[:H].sum()):.4f}, "
              f"token share {float(f[:H].sum()):.4f}, lambda of the top row {float(lam[0]):.1f}, of row 64 {float(lam[min(63, c.numel() - 1)]):.1f}", flush=True)
    del g, vocab
    torch.cuda.empty_cache()
