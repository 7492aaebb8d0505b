"""Context for the SGNS roofline: how fast can this GPU gather / update random 512-B (dim 128 fp32)
rows out of a matrix of a given size, with a plain torch gather / index_add (run on the GPU box).
python scripts/random_row_ceiling.py"""
import time, torch
dev = "cuda"
dim = 128
n_idx = 1 << 24  # 16 M rows = 8.6 GB gathered
for rows in (1 << 20, 10_000_000, 100_000_000):
    x = torch.zeros(rows, dim, device=dev)
    idx = torch.randint(0, rows, (n_idx,), device=dev)
    out = torch.empty(n_idx, dim, device=dev)
    src = torch.ones(n_idx, dim, device=dev)
    for name, fn, bytes_moved in (
        ("gather   rows -> dense", lambda: torch.index_select(x, 0, idx, out=out), 2 * n_idx * dim * 4),
        ("scatter  dense -> rows", lambda: x.index_copy_(0, idx, src), 2 * n_idx * dim * 4),
        ("rmw      rows += dense", lambda: x.index_add_(0, idx, src), 3 * n_idx * dim * 4),
    ):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.time(); fn(); torch.cuda.synchronize(); best = min(best, time.time() - t)
        print(f"matrix {rows*dim*4/1e9:6.1f} GB  {name}: {best*1e3:7.1f} ms  {bytes_moved/best/1e12:.2f} TB/s (random side: {n_idx*dim*4/best/1e12:.2f} TB/s)", flush=True)
    del x, idx, out, src
    torch.cuda.empty_cache()
