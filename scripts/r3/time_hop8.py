"""p = q = 1 walks with the 16-byte hop table, the 8-byte one (rows as in the CSR) and the padded
8-byte one, on a BASELINE graph: python time_hop8.py cfg3|cfg4"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import randomwalk as rw, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = 10_000_000 if which == "cfg3" else 100_000_000
g = synthetic.chung_lu(n, (10 if which == "cfg3" else 5) * n, seed=42, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)[: 1 << 20]
deg = g.degrees()
out = None


def run(tag, **kw):
    global out
    for _ in range(2):
        w = rw.walk(g, start, 10, 80, 1.0, 1.0, 42, check=False, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        w = rw.walk(g, start, 10, 80, 1.0, 1.0, 42, check=False, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    same = True if out is None else bool(torch.equal(out[0], w[0]))
    out = w if out is None else out
    print(f"{tag}: {int(w[1].sum()) * 80 / dt / 1e9:6.2f} G steps/s ({dt * 1e3:.2f} ms), same walks: {same}", flush=True)


run("16-byte hop table", use_hops8=False)
for shift in (0, 3):
    g.build_hops8(force=True, align_shift=shift)
    if g.hops8 is None:
        print(f"8-byte table, shift {shift}: does not fit"); continue
    cb, rb = g.hops8_bits
    esc = (1 << (64 - cb - rb)) - 1
    share = float(deg[deg >= esc].sum()) / g.n_edges
    print(f"8-byte table, shift {shift}: {cb}+{rb}+{64 - cb - rb} bits, escape at degree {esc}: "
          f"{int((deg >= esc).sum())} rows, {share:.3f} of the edge endpoints, table {g.hops8.numel() * 8 / 1e9:.2f} GB")
    run(f"8-byte hop table, shift {shift}")
