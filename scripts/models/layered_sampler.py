"""Model of the layered sampler of fast mode (csrc/n2v_walk_fast.hip, kClassFirst): for random class
configurations of a row (n slots, nR return slots in a run, nM shared slots at listed positions, the
rest "other") and every ordering of the three weights 1/p, 1, 1/q (ties included), the probability
the sampler gives a slot -- summed over the layers that contain it, (layer mass / total) / |set|, a
layer with "other" slots being a uniform draw over the row repeated inside the layer until it lands
in the set -- equals the reference's weight / sum of weights (randomwalk.py:219-231 followed by the
normalisation of the alias table), in exact rational arithmetic.  Also simulates the kernel's
selection (layer from uc in [0, total), slot by index or by rejection) with a deterministic sweep."""
import os
import random
import sys
from fractions import Fraction as F

trials = int(os.environ.get("N2V_MODEL_TRIALS", "2000"))
rng = random.Random(5)
bad = total = 0
PQ = [F(1, 4), F(1, 2), F(7, 10), F(1), F(13, 10), F(2), F(3), F(4)]
for _ in range(trials):
    n = rng.choice([1, 2, 3, 5, 8, 17, 64, 65, 200])
    nR = rng.choice([0, 0, 1, 1, 2, 3])
    nR = min(nR, n)
    rpos = rng.randint(0, n - nR)
    free = [j for j in range(n) if not (rpos <= j < rpos + nR)]
    nM = rng.randint(0, min(len(free), rng.choice([0, 1, 3, 40])))
    shared = sorted(rng.sample(free, nM))
    p, q = rng.choice(PQ), rng.choice(PQ)
    if q == 1:
        shared, nM = [], 0  # the kernel counts shared slots as "other" when q == 1
    w = {"R": 1 / p, "M": F(1), "O": 1 / q}
    cls = ["R" if rpos <= j < rpos + nR else ("M" if j in set(shared) else "O") for j in range(n)]
    weight = [w[c] for c in cls]
    tot_w = sum(weight)
    # the kernel's order: bubble sort of (1/p, 1, 1/q) with the bits (R, M, O), stable on ties
    wc, bc = [w["R"], w["M"], w["O"]], ["R", "M", "O"]
    for a in range(2):
        for b in range(2 - a):
            if wc[b] > wc[b + 1]:
                wc[b], wc[b + 1] = wc[b + 1], wc[b]
                bc[b], bc[b + 1] = bc[b + 1], bc[b]
    cnt = {"R": nR, "M": nM, "O": n - nR - nM}
    layers = [({"R", "M", "O"}, F(n) * wc[0]),
              ({"R", "M", "O"} - {bc[0]}, F(n - cnt[bc[0]]) * (wc[1] - wc[0])),
              ({bc[2]}, F(cnt[bc[2]]) * (wc[2] - wc[1]))]
    m_tot = sum(m for _, m in layers)
    prob = [F(0)] * n
    for S, m in layers:
        members = [j for j in range(n) if cls[j] in S]
        if m == 0:
            continue
        if not members:
            bad += 1  # a layer with mass but no slot: must never happen
            continue
        for j in members:
            prob[j] += m / m_tot / len(members)
    ok = m_tot == tot_w and all(prob[j] == weight[j] / tot_w for j in range(n))
    # the by-index draw of a set without "other" slots covers the return run, then the list
    for S, m in layers:
        if "O" not in S and m > 0:
            cr, cm = (nR if "R" in S else 0), (nM if "M" in S else 0)
            wl = m / (cr + cm)
            seen = []
            for kk in range(cr + cm):
                ul = wl * kk + wl / 2  # a point of the kk-th cell of the layer's mass
                k2 = min(int(ul / wl), cr + cm - 1)
                seen.append(rpos + k2 if k2 < cr else shared[k2 - cr])
            ok = ok and sorted(seen) == sorted(j for j in range(n) if cls[j] in S)
    total += 1
    bad += 0 if ok else 1
print("total", total, "rows,", bad, "bad")
sys.exit(1 if bad else 0)
