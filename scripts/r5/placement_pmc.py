"""The placement effect of the walk kernels, with counters.

One process; a fixed sequence of placements of the ranked table (3 GB) and the output buffer (3.4 GB):
the table as built / cloned into fresh allocations, the output buffer fresh or moved, and both carved
out of ONE allocation ("arena").  Every placement: 1 warm-up + REPS timed launches of the ranked walk
kernel (p = q = 1, 2^20 start vertices x 10 x 80), timed by HIP events; the virtual addresses are logged.
Under `rocprofv3 --pmc ...` the dispatches of walk_uniform_kernel come in this order (1 + REPS per
placement), so scripts/r5/placement_condense.py can put the counters of a placement beside its time.
  python scripts/r5/placement_pmc.py            # timings only
"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
REPS = int(os.environ.get("REPS", 3))
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked()
B, W, L = 1 << 20, 10, 80
valid = torch.empty(B * W, dtype=torch.uint8, device="cuda")
n_words = B * W * (L + 1)

def run(table, out, tag):
    g.rank_hops = table
    def step(k):
        rw.walk(g, start[k * B:(k + 1) * B], W, L, 1.0, 1.0, 42, out=(out, valid), check=False, rank_ids=True)
    step(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(1, 1 + REPS): step(k)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / REPS
    ta, oa = table.data_ptr(), out.data_ptr()
    print(f"PLACEMENT {tag}: {ms:.2f} ms  table {ta:#x} (mod 2M {ta % (1 << 21):#x}, mod 1G {ta % (1 << 30):#x}) "
          f"out {oa:#x} (mod 2M {oa % (1 << 21):#x}, mod 1G {oa % (1 << 30):#x})  distance {abs(ta - oa) / 2**30:.3f} GiB", flush=True)
    return ms

t0 = g.rank_hops
keep = [t0]
def fresh_out():
    o = torch.empty((B * W, L + 1), dtype=torch.int32, device="cuda"); keep.append(o); return o
def clone_table():
    t = t0.clone(); keep.append(t); return t
o0 = fresh_out()
run(t0, o0, "A table as built, out 0")
t1 = clone_table()
run(t1, o0, "B clone 1, out 0")
o1 = fresh_out()
run(t1, o1, "C clone 1, out 1")
t2 = clone_table()
run(t2, o1, "D clone 2, out 1")
o2 = fresh_out()
run(t2, o2, "E clone 2, out 2")
run(t0, o2, "F table as built, out 2")
# both buffers inside ONE allocation: table first, output behind it at the next 2 MB boundary
for rep in range(3):
    tb = t0.numel() * 4
    off = (tb + (1 << 21) - 1) >> 21 << 21
    arena = torch.empty(off + n_words * 4 + (1 << 21), dtype=torch.uint8, device="cuda"); keep.append(arena)
    base = (-arena.data_ptr()) % (1 << 21)  # 2 MB-aligned start inside the arena
    ta = arena[base:base + tb].view(torch.int32); ta.copy_(t0)
    oa = arena[base + off:base + off + n_words * 4].view(torch.int32).view(B * W, L + 1)
    run(ta, oa, f"G{rep} one allocation: table, then out")
    # and with a gap of 1 GiB between them inside a larger allocation
    arena2 = torch.empty(off + n_words * 4 + (1 << 30) + (1 << 21), dtype=torch.uint8, device="cuda"); keep.append(arena2)
    base = (-arena2.data_ptr()) % (1 << 21)
    tb2 = arena2[base:base + tb].view(torch.int32); tb2.copy_(t0)
    ob2 = arena2[base + off + (1 << 30):base + off + (1 << 30) + n_words * 4].view(torch.int32).view(B * W, L + 1)
    run(tb2, ob2, f"H{rep} one allocation: table, 1 GiB gap, out")
print("allocated GB", torch.cuda.memory_allocated() / 1e9, flush=True)
