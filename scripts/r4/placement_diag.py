"""what distinguishes an output buffer on which the ranked walk runs in 16.8 ms from one on which it takes
19.6 ms?  Finds a fast and a slow buffer with the walk kernel, then times plain kernels on both."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import bench
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked()
def ev(fn, reps=4):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
leg = bench.WalkLeg(torch, rw, g, start, 10, 80, 1.0, 1.0, "exact", 1 << 20, 0, 1, rank_ids=True)
bufs = [leg.walks] + [torch.empty_like(leg.walks) for _ in range(7)]
times = []
for b in bufs:
    leg.walks = b
    k = [0]
    def step():
        leg.step(2 + k[0] % 6); k[0] += 1
    times.append(ev(step, 6))
print("walk ms per output buffer:", [round(x, 2) for x in times], flush=True)
print("addresses:", [hex(b.data_ptr()) for b in bufs], flush=True)
fast, slow = bufs[times.index(min(times))], bufs[times.index(max(times))]
idx = torch.randint(0, fast.numel() // 16, (1 << 26,), device="cuda")
src16 = torch.zeros((1 << 26, 16), dtype=torch.int32, device="cuda")
for name, b in (("fast", fast), ("slow", slow)):
    flat = b.view(-1)
    rows16 = flat[: (flat.numel() // 16) * 16].view(-1, 16)
    print(name, f"fill {ev(lambda: flat.fill_(1)):.2f} ms; sum {ev(lambda: flat.sum()):.2f} ms; "
          f"random 64-byte row gather (67 M rows) {ev(lambda: rows16[idx]):.2f} ms; "
          f"random 64-byte row scatter {ev(lambda: rows16.index_copy_(0, idx, src16)):.2f} ms", flush=True)
