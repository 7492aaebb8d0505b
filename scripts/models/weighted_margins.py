"""Model of walk_weighted_margin_kernel (csrc/n2v_walk_wlanes.hip): the draw of ONE slot of the table
generate_alias_tables builds (randomwalk.py:157-190) decided from sums over the row with margins, against
the table itself (oracle/n2v_oracle: the reference's loop, fp64).  Every DECIDED draw must be the table's;
the share left undecided is printed.  As in the kernel a draw the general margins leave undecided gets a second
chance with the row added up in the reference's own order (the margins of an exact sum).
  python scripts/models/weighted_margins.py [rows] [seed]      (also run by tests/test_closed_form_models.py)
TEST INFRASTRUCTURE (like oracle/, which it checks against): nothing in node2vec_amd/ imports it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import n2v_oracle as O  # noqa: E402

EPS = 2.220446049250313e-16
UNDECIDED = -2


EXACT_DELTA = 8.0 * EPS


def general_margins(n, kfac=1.0):
    """(delta, M) for any weights and any p, q: probs are known to delta (relative), sums of up to n of them to M"""
    nn = float(n)
    return kfac * (2.0 * nn + 16.0) * EPS, kfac * 16.0 * nn * nn * EPS


def class_factor_spread(factors):
    """kfac of general_margins when the row sum is taken from three terms (stored row sum, shared, return)"""
    cq, c1, cp = factors
    return 1.0 + 2.0 * max(cq, cp, 1.0) / min(cq, cp, 1.0)


def exact_sum_margin(n, max_prob, tot_d):
    """M when the row sum is the reference's bit for bit (+ 64: a block summary is the difference of two sums of
    up to 256 weights)"""
    nn = float(n)
    return 8.0 * EPS * (nn * (max_prob + 12.0 + 64.0) + (nn / 256.0 + 16.0) * (4.0 * tot_d + 4.0))


def crossing(vals, target):
    """first slot from the TOP whose running sum (descending position) reaches target: (slot, before, at, whole)"""
    run = np.cumsum(vals[::-1])  # run[t] = sum of vals[n - 1 - t ..]
    hit = np.nonzero(run >= target)[0]
    if hit.size == 0:
        return -1, 0.0, 0.0, float(run[-1]) if run.size else 0.0
    t = int(hit[0])
    return len(vals) - 1 - t, float(run[t] - vals[len(vals) - 1 - t]), float(run[t]), float(run[-1])


def margin_draw(b, pick, r2, rng=None, w=None, cls=None, factors=None, exact_total=False, sequential=False):
    """b: biased weights (fp64, >= 0); returns the slot sampling_from_alias returns, or UNDECIDED.  With the
    stored weights w, the classes and the three factors the row sum is taken as the kernel takes it: from the
    sum of the stored weights and the sums over the shared and the return slots"""
    n = len(b)
    order = np.arange(n) if rng is None else rng.permutation(n)  # (the kernel adds in another order)
    if not np.all(b >= 0.0):
        return UNDECIDED
    kfac = 1.0
    if sequential:  # the second chance: the row added up in the reference's own order -- its sum, bit for bit
        total = float(np.cumsum(b)[-1])
        exact_total = True
    elif w is None:
        total = float(np.sum(b[order]))
    else:
        cq, c1, cp = factors  # other, shared, return
        row_sum = float(np.sum(w[order]))
        ss = float(np.sum(w[cls == 1]))
        sr = float(np.sum(w[cls == 2]))
        total = cq * row_sum + (1.0 - cq) * ss + (cp - cq) * sr
        kfac = class_factor_spread(factors)
    if not (total > 0.0):
        return UNDECIDED
    nn = float(n)
    inv = nn / total
    delta, M = general_margins(n, kfac)
    if exact_total:
        # every addend a multiple of one power of two and the sum below 2^52 of them: the reference's
        # left-to-right sum rounds nowhere and is THIS sum; what is left are the roundings of the loop
        # (2 per pairing, at most max(probs) + 1 in size) and of the sums taken here
        delta = EXACT_DELTA
    p_pick = b[pick] * inv
    under = p_pick < 1.0 - 2.0 * delta
    if not under and not (p_pick > 1.0 + 2.0 * delta):
        return UNDECIDED
    if under:
        if r2 < p_pick * (1.0 - delta):
            return pick
        if not (r2 > p_pick * (1.0 + delta)):
            return UNDECIDED
    x = b * inv - 1.0
    d = np.maximum(-x, 0.0)
    e = x + d
    pr = b * inv
    # running sums from the bottom, as the kernel keeps them (per lane, then over the wave)
    tot_d, tot_x = float(np.sum(d[order])), float(np.sum(x[order]))
    below_d, below_x = float(np.sum(d[:pick])), float(np.sum(x[:pick]))
    if w is not None and n >= 300:
        # a long row: the sums as wm_hub_sums takes them -- per block of 256 slots the sorted weights and their
        # prefix sums give the block's d and x AS IF every slot were "other" (a count below the threshold and a
        # prefix sum), the shared and return slots are corrected one by one
        cq, c1, cp = factors
        f_other = cq * inv
        fac = np.where(cls == 2, cp, np.where(cls == 1, c1, cq)) * inv
        blk_d, blk_x = [], []
        for c0 in range(0, n, 256):
            srt = np.sort(w[c0:c0 + 256])
            pre = np.concatenate([[0.0], np.cumsum(srt)])
            k = int(np.sum(srt * f_other - 1.0 < 0.0))
            blk_d.append(k - f_other * pre[k])
            blk_x.append(f_other * pre[len(srt)] - len(srt))
        xo = w * f_other - 1.0
        xs = w * fac - 1.0
        corr_x = np.where(cls != 0, xs - xo, 0.0)
        corr_d = np.where(cls != 0, np.maximum(-xs, 0.0) - np.maximum(-xo, 0.0), 0.0)
        tot_d = float(np.sum(blk_d) + np.sum(corr_d))
        tot_x = float(np.sum(blk_x) + np.sum(corr_x))
        bp = pick // 256
        part = slice(bp * 256, pick)
        below_d = float(np.sum(blk_d[:bp]) + np.sum(np.maximum(-xo[part], 0.0)) + np.sum(corr_d[:pick]))
        below_x = float(np.sum(blk_x[:bp]) + np.sum(xo[part]) + np.sum(corr_x[:pick]))
    if exact_total:
        M = exact_sum_margin(n, float(np.max(b)) * inv, tot_d)

    def next_over_below(top):  # the first slot below `top` that is overfull, every slot skipped underfull
        for t in range(top - 1, -1, -1):
            if pr[t] > 1.0 + 2.0 * delta:
                return t
            if not (pr[t] < 1.0 - 2.0 * delta):
                return UNDECIDED
        return UNDECIDED

    if under:
        d_above = tot_d - below_d - (-(p_pick - 1.0))
        if not (d_above > M):
            t = next_over_below(n)
            if t < 0:
                return UNDECIDED
            return t if pr[t] - 1.0 >= 3.0 * M else UNDECIDED
        k, before, at, _ = crossing(e, d_above)
        if k < 0 or not (before <= d_above - M) or not (at >= d_above + M):
            return UNDECIDED
        return k
    e_from = (tot_x + tot_d) - (below_x + below_d)
    if tot_d <= e_from - M:
        return pick  # never demoted
    if tot_d <= e_from + M:  # demoted, if at all, with probs >= 1 - 2 M
        return pick if r2 < 1.0 - 3.0 * M else UNDECIDED
    j, before, at, _ = crossing(d, e_from)
    if j < 0 or not (before <= e_from - M) or not (at >= e_from + M):
        return UNDECIDED
    resid = 1.0 + e_from - at
    if r2 < resid - M:
        return pick
    if not (r2 > resid + M):
        return UNDECIDED
    return next_over_below(pick)


def exact_draw(b, pick, r2):
    alias, probs = O.alias_tables(b)
    return pick if r2 < probs[pick] else int(alias[pick])


def rows(rng, count):
    for i in range(count):
        kind = i % 7
        n = int(rng.integers(1, 400)) if kind != 5 else int(rng.integers(400, 6000))
        if kind == 0:
            w = rng.random(n)
        elif kind == 1:
            w = rng.random(n).astype(np.float32).astype(np.float64)
        elif kind == 2:  # heavy tail: one slot carries most of the row
            w = rng.pareto(1.1, n) + 1e-3
        elif kind == 3:  # few distinct values: sums tie exactly (undecided often, never wrong)
            w = rng.integers(1, 4, n).astype(np.float64)
        elif kind == 4:  # probs close to 1.0
            w = 1.0 + rng.normal(0, 1e-13, n) * rng.integers(0, 2, n)
        elif kind == 6:  # sums that tie up to a few ulps
            w = rng.integers(1, 4, n).astype(np.float64) * (1.0 + rng.integers(-2, 3, n) * 2.0 ** -52)
        else:
            w = rng.random(n) * rng.choice([1.0, 1e-3, 1e3], n)
        # the p / q classes of a step: a return slot, some shared slots, the rest "other"
        p, q = rng.choice([0.25, 0.5, 0.7, 1.0, 2.0, 3.0, 4.0], 2)
        cls = (rng.random(n) < 0.3).astype(int)
        cls[int(rng.integers(0, n))] = 2
        b = np.where(cls == 2, w / p, np.where(cls == 1, w, w / q))
        exact = False
        if kind in (1, 3) and p in (0.25, 0.5, 1.0, 2.0, 4.0) and q in (0.25, 0.5, 1.0, 2.0, 4.0) and np.all(w > 0):
            # fp32 weights (24 bits), factors powers of two: the grid of the addends and the room above it
            grid = 2.0 ** (np.frexp(w)[1].min() - 24) * min(1.0 / p, 1.0 / q, 1.0)
            exact = n * float(np.max(b)) / grid < 2.0 ** 48
        yield b, w, cls, (1.0 / q, 1.0, 1.0 / p), exact


def main(count=3000, seed=1):
    rng = np.random.default_rng(seed)
    draws = undecided = second = 0
    per_kind = {}
    for i, (b, w, cls, factors, exact) in enumerate(rows(rng, count)):
        n = len(b)
        alias, probs = O.alias_tables(b)
        picks = np.arange(n) if n <= 64 else rng.integers(0, n, 64)
        for pick in picks:
            # uniforms of the grid the kernels use, and adversarial ones right at the table's threshold
            cand = [float(rng.integers(0, 2 ** 32)) / 2 ** 32, float(probs[pick]),
                    float(np.nextafter(probs[pick], 0.0)), float(np.nextafter(probs[pick], 2.0))]
            cand += [float(probs[pick]) + sgn * 10.0 ** -int(rng.integers(6, 15)) for sgn in (-1.0, 1.0)]
            # the kernels' uniforms are u / 2^32: the grid points around the table's threshold
            base = np.floor(float(probs[pick]) * 4294967296.0)
            cand += [float(base + k) / 4294967296.0 for k in (-1, 0, 1, 2)]
            for which, r2 in enumerate(cand):
                if not (0.0 <= r2 < 1.0):
                    continue
                want = int(pick) if r2 < probs[pick] else int(alias[pick])
                got = margin_draw(b, int(pick), r2, rng, w, cls, factors, exact)
                if got == UNDECIDED and not exact:  # the second chance of the kernel
                    got = margin_draw(b, int(pick), r2, rng, w, cls, factors, sequential=True)
                    second += 1
                draws += 1
                k = per_kind.setdefault(i % 7, [0, 0, 0, 0])  # random r2: draws, undecided; at the threshold: same
                k[0 if which == 0 else 2] += 1
                if got == UNDECIDED:
                    undecided += 1
                    k[1 if which == 0 else 3] += 1
                elif got != want:
                    print(f"WRONG: row {i} (n = {n}), pick {pick}, r2 {r2!r}: decided {got}, the table says {want}")
                    return 1
    print(f"{count} rows, {draws} draws: 0 wrong, {second} had the second chance (the row sum in the reference's order), "
          f"{undecided} undecided ({undecided / max(draws, 1):.4f}); "
          f"by kind of row (uniform r2: draws, undecided; r2 at the table's threshold: draws, undecided): {per_kind}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
