"""Rows built so that the INTERNAL comparisons of the "decided, not replayed" kernels sit at chosen multiples of
their margins -- where random fuzz never lands.

Two decision procedures are attacked, through their Python models (same formulas, line by line, as the kernels):
  near      n2v_unit_near.h       (near_forms.near_step / near_step_exact): unit weights, 1/p or 1/q not dyadic;
                                  margins 5e-15 n (n + 8) vmax (counts) and 2e-14 n (vmax + 1) (after the exact sum)
  weighted  n2v_walk_wlanes.hip   (weighted_margins.margin_draw): arbitrary weights; general margins
                                  kfac 16 n^2 2^-52 and the margins of an exact row sum (linear in n)

How a row is placed.  The reference's loop (generate_alias_tables, randomwalk.py:172-189) is run in EXACT rational
arithmetic on the row's fp64 values (all of them dyadic rationals: scaled to integers, `probs[i] < 1.0` is
`n W_i < sum W`; no rounding anywhere).  Every iteration yields the exact value  P - 1  of the slot it pushes back:
the quantity  X_i - k D  /  E_k - D_j  whose sign the closed forms decide.  One knob of the row -- q for unit
weights (b_other = 1 / q), the stored weight of one underfull slot for weighted rows (on the fp32 grid when the
row is to qualify for the exact-sum margins) -- is then moved by Newton steps on the exact values until the
iteration nearest to a tie sits at  t x margin,  t in +-{0.25, 0.5, 0.9, 1.1, 2, 10}; `pick` against 1.0 is placed
the same way on probs[pick] - 1 against 2 delta.  The achieved multiple is recomputed exactly and reported.

What is asserted.  The truth is the reference's own fp64 loop (oracle / ref_tables).  For the slots of the placed
iteration, their neighbours on the stacks and a few others, with r2 on the u / 2^32 grid around probs[pick], at
it, one ulp beside it and random:  decided  =>  the reference's draw.  Declining is always allowed; the share is
printed per multiple (a correct procedure declines inside +-1 and should decide outside).

  python scripts/models/margin_adversary.py near|weighted [n_max] [rows per (n, t)] [seed]
TEST INFRASTRUCTURE (tests/test_closed_form_models.py, tests/test_margin_adversary_gpu.py import it): nothing
in node2vec_amd/ does."""
import math
import os
import random
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for _p in (HERE, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import near_forms as NF  # noqa: E402
import weighted_margins as WM  # noqa: E402
from oracle import n2v_oracle as O  # noqa: E402

TARGETS = (0.01, 0.05, 0.25, 0.5, 0.9, 1.1, 2.0, 10.0)  # (0.01, 0.05: where fp64 rounding really flips the reference)
SIZES = (8, 30, 100, 1000, 10_000, 100_000)


def targets_for(n):
    return TARGETS if n < 10_000 else (0.25, 0.9, 1.1, 2.0)  # (a placement on 10^5 slots is seconds of exact arithmetic)


def signs_for(n, rng):
    return (1.0, -1.0) if n < 10_000 else (rng.choice([1.0, -1.0]),)


# ---- the reference's loop in exact arithmetic ----------------------------------------------------------------
def exact_events(b):
    """generate_alias_tables on the fp64 values b, no rounding: (alias, A, T, events) with probs[i] = A[i] / T
    exactly at the end and events = [(over slot, under slot, (P - 1) T)] per iteration, P the value pushed back"""
    n = len(b)
    ratios = [float(x).as_integer_ratio() for x in b]
    K = max(d for _, d in ratios)  # denominators are powers of two
    W = [num * (K // den) for num, den in ratios]
    T = sum(W)
    A = [x * n for x in W]
    under = [i for i in range(n) if A[i] < T]
    over = [i for i in range(n) if not A[i] < T]
    alias = [0] * n
    events = []
    while under and over:
        u, o = under.pop(), over.pop()
        alias[u] = o
        A[o] = A[o] + A[u] - T
        events.append((o, u, A[o] - T))
        (under if A[o] < T else over).append(o)
    return alias, A, T, events


def _ratio(num, den):
    return float(Fraction(num, den))


def place(make_row, knob0, step_knob, margin_of, want, rounds=40):
    """move the knob until the iteration nearest to a tie has (P - 1) = want x margin.  make_row(knob) -> b;
    step_knob(knob, delta) -> the representable knob nearest to knob + delta (None: no further);
    margin_of(b) -> the margin.  Returns (knob, achieved multiple, (over, under)) or None."""
    def events_of(k):
        _, _, T, ev = exact_events(make_row(k))
        return T, ev

    k = knob0
    T, ev = events_of(k)
    # the last iteration ends at exactly 1.0 by mass balance: not a comparison anybody makes
    ev = ev[:-1]
    if not ev:
        return None
    near = min(range(len(ev)), key=lambda i: abs(ev[i][2]))
    ident = ev[near][:2]
    best = None
    for _ in range(rounds):
        T, ev = events_of(k)
        hit = [e for e in ev[:-1] if e[:2] == ident]
        if not hit:  # the staircase changed before the tie was reached: take the tie that is nearest now
            if not ev[:-1]:
                return best
            e = min(ev[:-1], key=lambda x: abs(x[2]))
            ident = e[:2]
        else:
            e = hit[0]
        g = _ratio(e[2], T)
        M = margin_of(make_row(k))
        got = g / M
        if best is None or abs(got - want) < abs(best[1] - want):
            best = (k, got, ident)
        if abs(got - want) <= 0.04 * abs(want):
            break
        # slope by a finite difference on the exact values
        h = abs(k) * 1e-9 + 1e-300
        k2 = step_knob(k, h)
        if k2 is None or k2 == k:
            break
        T2, ev2 = events_of(k2)
        hit2 = [x for x in ev2[:-1] if x[:2] == ident]
        if not hit2:
            break
        slope = (_ratio(hit2[0][2], T2) - g) / (k2 - k)
        if slope == 0.0 or not math.isfinite(slope):
            break
        nk = step_knob(k, (want * M - g) / slope)
        if nk is None or nk == k:
            break
        k = nk
    return best


# ---- near: unit weights, class values that are not dyadic ------------------------------------------------------
def near_row_values(cls, p, q):
    return {'R': 1.0 / p, 'M': 1.0, 'O': 1.0 / q}


def near_margins(n, cls, b):
    G = NF.Row(n, cls)
    approx = (G.nR * b['R'] + G.nM * b['M'] + G.nO * b['O']) / float(n)
    vmax = max([b[c] / approx for c, k in (('R', G.nR), ('M', G.nM), ('O', G.nO)) if k > 0])
    return 5e-15 * float(n) * (float(n) + 8.0) * vmax, 2e-14 * float(n) * (vmax + 1.0)


def near_truth(cls, b):
    return NF.ref_tables([b[c] for c in cls])


def near_decide(n, cls, pick, r2, b):
    """as the kernel: the forms on the counts (quadratic margin), then on the exact row sum (linear margin); None =
    the replay.  Returns (slot or None, stage that decided)"""
    got = NF.near_step(n, cls, pick, r2, b)
    if got is not None:
        return got, 1
    got = NF.near_step_exact(n, cls, pick, r2, b)
    return got, (2 if got is not None else 0)


def r2_candidates(prob, rng):
    u0 = int(math.floor(min(max(prob, 0.0), 1.0) * 2.0 ** 32))
    cand = [min(max(u0 + d, 0), 2 ** 32 - 1) / 2 ** 32 for d in range(-2, 3)]
    cand += [prob, math.nextafter(prob, 0.0), math.nextafter(prob, 2.0)]
    cand += [rng.getrandbits(32) / 2 ** 32 for _ in range(2)]
    return [r for r in cand if 0.0 <= r < 1.0]


def core_r2(prob):
    """uniforms well away from probs[pick] on both sides (on the u / 2^32 grid)"""
    out = []
    for r in (0.5 * prob, 0.5 * (min(prob, 1.0) + 1.0)):
        r = math.floor(r * 2.0 ** 32) / 2.0 ** 32
        if 0.0 <= r < 1.0 and abs(r - prob) > 1e-6:
            out.append(r)
    return out


def near_arrangement(n, cls, b):
    G = NF.Row(n, cls)
    avg = sum(b[c] for c in cls) / n
    uR, uM, uO = b['R'] < avg, b['M'] < avg, b['O'] < avg
    nR, nM, nO = G.nR, G.nM, G.nO
    if uO and not (nR and uR) and not (nM and uM): return 1
    if not uO and nO > 0 and (not nR or uR) and (not nM or uM): return 2
    if uO and nR and uR and nM and not uM: return 3
    if not uO and nO > 0 and nR and not uR and nM and uM: return 4
    if uO and nR and not uR and nM and uM: return 5
    return 0


# (p, q) that produce each of the five arrangements on rows with a return run and a shared list
NEAR_PQ = {1: [(0.7, 3.0), (1.3, 2.5), (0.6, 7.0)], 2: [(3.0, 0.7), (5.0, 0.2), (1.5, 0.6)],
           3: [(3.0, 1.5), (6.0, 2.5), (37.5, 1.2)], 4: [(0.15, 0.45), (0.1, 0.6), (0.2, 0.35)],
           5: [(0.05, 1.3), (0.1, 1.2), (1.0 / 7.0, 1.5)]}


def near_random_cls(n, rng, arr):
    cls = ['O'] * n
    nR = rng.choice([1, 1, 2, 3]) if n > 4 else 1
    rp = rng.randint(0, n - nR)
    for k in range(nR):
        cls[rp + k] = 'R'
    dens = rng.choice([0.03, 0.1, 0.3, 0.5])
    if arr == 5:
        dens = rng.choice([0.3, 0.6])  # (shared slots under the average need a heavy return run)
    for i in range(n):
        if cls[i] == 'O' and rng.random() < dens:
            cls[i] = 'M'
    if 'M' not in cls:
        free = [i for i in range(n) if cls[i] == 'O']
        if free:
            cls[rng.choice(free)] = 'M'
    return cls


def attack_near(n_max, per, seed, out=print, collect=None):
    rng = random.Random(seed)
    table = {}  # (stage margin, t) -> [rows, draws, declined, wrong]
    wrong = 0
    for n in [x for x in SIZES if x <= n_max]:
        reps = per if n <= 1000 else max(1, per // 4)
        for arr in (1, 2, 3, 4, 5):
            for which in (0, 1):  # the margin the multiple refers to: counts (n^2), exact sum (n)
                for t in targets_for(n):
                    for sign in signs_for(n, rng):
                        for _ in range(reps):
                            p, q0 = rng.choice(NEAR_PQ[arr])
                            cls = near_random_cls(n, rng, arr)
                            if near_arrangement(n, cls, near_row_values(cls, p, q0)) != arr:
                                continue
                            make = lambda q: [near_row_values(cls, p, q)[c] for c in cls]  # noqa: E731
                            marg = lambda bb: near_margins(n, cls, {'R': 1.0 / p, 'M': 1.0, 'O': bb[cls.index('O')]  # noqa: E731
                                                                    if 'O' in cls else 1.0})[which]
                            step = lambda k, d: float(np.float64(k) + np.float64(d))  # noqa: E731
                            got = place(make, q0 * (1.0 + rng.uniform(-0.02, 0.02)), step, marg, sign * t)
                            if got is None:
                                continue
                            q, mult, (o, u) = got
                            if not (0.2 * t <= abs(mult) <= 5.0 * t):
                                continue
                            b = near_row_values(cls, p, q)
                            if near_arrangement(n, cls, b) != arr:
                                continue
                            if collect is not None:
                                collect.append(dict(n=n, cls=cls, p=p, q=q, slots=(o, u), multiple=mult, arr=arr,
                                                    margin="counts" if which == 0 else "exact-sum"))
                            alias, probs = near_truth(cls, b)
                            key = ("n^2" if which == 0 else "n", t)
                            row = table.setdefault(key, [0, 0, 0, 0, 0, 0, 0])
                            row[0] += 1
                            xalias, xA, xT, _ = exact_events([b[c] for c in cls])
                            # the draws the placed iteration decides, r2 far from probs[pick]: who absorbed the
                            # underfull slot, where the overfull one was demoted and who absorbed IT
                            for pick in (o, u):
                                for r2 in core_r2(probs[pick]):
                                    p0 = b[cls[pick]] / (sum(b[c] for c in cls) / n)
                                    if p0 < 1.0 and r2 < p0:
                                        continue
                                    want = pick if r2 < probs[pick] else alias[pick]
                                    dec, _stage = near_decide(n, cls, pick, r2, b)
                                    row[4] += 1
                                    row[6] += want != (pick if Fraction(r2) < Fraction(xA[pick], xT) else xalias[pick])
                                    if dec is None:
                                        row[5] += 1
                                    elif dec != want:
                                        row[3] += 1
                                        wrong += 1
                                        out(f"WRONG near (core): n {n} arr {arr} p {p!r} q {q!r} pick {pick} r2 {r2!r}: "
                                            f"decided {dec}, the reference {want}; placed at {mult:.3g} x margin[{key[0]}]")
                            picks = {o, u, max(o - 1, 0), min(o + 1, n - 1), max(u - 1, 0), min(u + 1, n - 1)}
                            picks |= set(range(n)) if n <= 30 else {rng.randrange(n) for _ in range(4)}
                            avg = sum(b[c] for c in cls) / n
                            for pick in sorted(picks):
                                for r2 in r2_candidates(probs[pick], rng):
                                    p0 = b[cls[pick]] / avg
                                    if p0 < 1.0 and r2 < p0:
                                        continue  # the quick accept comes first in the kernel
                                    want = pick if r2 < probs[pick] else alias[pick]
                                    dec, _stage = near_decide(n, cls, pick, r2, b)
                                    row[1] += 1
                                    if dec is None:
                                        row[2] += 1
                                    elif dec != want:
                                        row[3] += 1
                                        wrong += 1
                                        if wrong < 8:
                                            out(f"WRONG near: n {n} arr {arr} p {p!r} q {q!r} pick {pick} r2 {r2!r}: "
                                                f"decided {dec}, the reference {want}; placed at {mult:.3g} x margin[{key[0]}]")
    out("near: multiple of the margin -> rows placed; draws on the two slots of the placed iteration with r2 far from "
        "probs[pick]: declined share; all draws (r2 at the threshold included): declined share; wrong")
    for key in sorted(table):
        r = table[key]
        out(f"  margin ~{key[0]:3s} x {key[1]:5.2f}: rows {r[0]:5d} placed slots {r[4]:6d} declined {r[5] / max(r[4], 1):.3f} | "
            f"all draws {r[1]:7d} declined {r[2] / max(r[1], 1):.3f} | wrong {r[3]} | the reference's fp64 loop differs from exact arithmetic on {r[6]} placed draws")
    tot = [sum(r[i] for r in table.values()) for i in range(4)]
    out(f"total {tot[1]} rows {tot[0]} declined {tot[2]} bad {tot[3]}")
    return tot


# ---- weighted rows --------------------------------------------------------------------------------------------
def f32(x):
    return float(np.float32(x))


def weighted_row(kind, n, rng):
    """(stored weights fp64 array, on the fp32 grid?).  fp32-representable weights qualify for the exact-sum margins
    when p and q are powers of two.  The LAST slots of an fp32 row are small weights: the underfull stack gives them
    up first, so every later iteration of the loop carries their deficit -- knobs whose fp32 steps (2^-24 of a small
    weight) are fine enough to address a margin that is linear in n."""
    if kind == "fp32":
        w = (rng.random(n) * 1.9 + 0.1).astype(np.float32).astype(np.float64)
        # how small: the row must stay eligible for the exact sum (n max(b) / grid < 2^48, grid = the last mantissa bit
        # of the smallest weight) and one fp32 step of the knob must move a sum by less than the margin (~2^-43 n): that
        # leaves about one binade, 2^(log2 n - 20)
        k = min(3, n // 3)
        e = math.floor(math.log2(n)) - 20
        w[n - k:] = (2.0 ** e * (0.5 + 0.5 * rng.random(k))).astype(np.float32).astype(np.float64)
        return w, True
    if kind == "fp64":
        return rng.random(n) * 1.9 + 0.1, False
    if kind == "decades":  # 24 decades
        return 10.0 ** rng.uniform(-12.0, 12.0, n), False
    if kind == "integers":  # sums that tie EXACTLY, everywhere: nothing to place, everything to decline or get right
        return rng.integers(1, 5, n).astype(np.float64), True
    raise ValueError(kind)


def weighted_classes(n, rng):
    cls = (rng.random(n) < rng.choice([0.05, 0.3])).astype(int)
    cls[int(rng.integers(0, n))] = 2
    return cls


def weighted_bias(w, cls, p, q):
    return np.where(cls == 2, w / p, np.where(cls == 1, w, w / q))


def weighted_M(b, factors, exact):
    n = len(b)
    if not exact:
        return WM.general_margins(n, WM.class_factor_spread(factors))[1]
    inv = n / float(np.sum(b))
    x = b * inv - 1.0
    return WM.exact_sum_margin(n, float(np.max(b)) * inv, float(np.sum(np.maximum(-x, 0.0))))


def weighted_decide(b, pick, r2, rng, w, cls, factors, exact):
    got = WM.margin_draw(b, pick, r2, rng, w, cls, factors, exact)
    if got == WM.UNDECIDED and not exact:
        got = WM.margin_draw(b, pick, r2, rng, w, cls, factors, sequential=True)
    return got


def exact_eligible(w, b, p, q):
    if not all(x in (0.25, 0.5, 1.0, 2.0, 4.0) for x in (p, q)) or not np.all(w > 0):
        return False
    grid = 2.0 ** (np.frexp(w)[1].min() - 24) * min(1.0 / p, 1.0 / q, 1.0)
    return len(b) * float(np.max(b)) / grid < 2.0 ** 48


def attack_weighted(n_max, per, seed, out=print, collect=None):
    rng = np.random.default_rng(seed)
    prng = random.Random(seed)
    table = {}
    wrong = 0
    unplaced = {}

    def check(row, b, w2, cls, factors, ex2, slots, alias, probs, n, label):
        nonlocal wrong
        xalias, xA, xT, _ = exact_events(b.tolist())
        for pick in slots:
            for r2 in core_r2(float(probs[pick])):
                want = int(pick) if r2 < probs[pick] else int(alias[pick])
                dec = weighted_decide(b, int(pick), r2, rng, w2, cls, factors, ex2)
                row[4] += 1
                row[6] += want != (int(pick) if Fraction(r2) < Fraction(xA[pick], xT) else xalias[pick])
                if dec == WM.UNDECIDED:
                    row[5] += 1
                elif dec != want:
                    row[3] += 1
                    wrong += 1
                    out(f"WRONG weighted (placed slot): {label} pick {pick} r2 {r2!r}: decided {dec}, the table {want}")
        picks = set(int(x) for x in slots)
        for x in list(picks):
            picks |= {max(x - 1, 0), min(x + 1, n - 1)}
        picks |= set(range(n)) if n <= 30 else {int(rng.integers(0, n)) for _ in range(4)}
        for pick in sorted(picks):
            for r2 in r2_candidates(float(probs[pick]), prng):
                want = pick if r2 < probs[pick] else int(alias[pick])
                dec = weighted_decide(b, pick, r2, rng, w2, cls, factors, ex2)
                row[1] += 1
                if dec == WM.UNDECIDED:
                    row[2] += 1
                elif dec != want:
                    row[3] += 1
                    wrong += 1
                    if wrong < 8:
                        out(f"WRONG weighted: {label} pick {pick} r2 {r2!r}: decided {dec}, the table {want}")

    for n in [x for x in SIZES if x <= n_max]:
        reps = per if n <= 1000 else max(1, per // 4)
        for kind in ("fp32", "fp64", "decades", "integers"):
            for where in ("crossing", "pick"):
                for t in targets_for(n):
                    for sign in signs_for(n, prng):
                        for _ in range(reps):
                            w, grid32 = weighted_row(kind, n, rng)
                            cls = weighted_classes(n, rng)
                            if grid32:
                                p, q = rng.choice([0.5, 2.0], 2)
                            else:
                                p, q = rng.choice([0.25, 0.5, 0.7, 2.0, 3.0, 4.0], 2)
                            p, q = float(p), float(q)
                            factors = (1.0 / q, 1.0, 1.0 / p)
                            b0 = weighted_bias(w, cls, p, q)
                            exact = grid32 and exact_eligible(w, b0, p, q)
                            label = f"n {n} kind {kind} {where} p {p} q {q}"
                            if kind == "integers":
                                # nothing to place: the running sums of such a row tie exactly all over the lattice
                                if where == "pick" or t != targets_for(n)[0]:
                                    continue
                                alias, probs = O.alias_tables(b0)
                                row = table.setdefault(("ties", "exact-sum" if exact else "general", 0.0), [0] * 7)
                                row[0] += 1
                                _, _, _, ev = exact_events(b0.tolist())
                                tied = [e[:2] for e in ev[:-1] if e[2] == 0][:4]
                                check(row, b0, w, cls, factors, exact, [x for e in tied for x in e], alias, probs, n, label)
                                continue
                            alias0, A0, T0, ev0 = exact_events(b0.tolist())
                            if len(ev0) < 3:
                                continue
                            step32 = lambda k, d: (f32(k + d) if f32(k + d) != k else  # noqa: E731
                                                   float(np.nextafter(np.float32(k), np.float32(k + math.copysign(1.0, d)))))
                            step64 = lambda k, d: (k + d if k + d != k else math.nextafter(k, k + math.copysign(1.0, d)))  # noqa: E731
                            step = step32 if grid32 else step64
                            fine = n - 1  # the slot the underfull stack gives up first (fp32 rows: a small weight)
                            under_first = A0[fine] < T0 and b0[fine] > 0

                            def make(x, at, w=w, cls=cls, p=p, q=q):
                                w2 = w.copy()
                                for i, v in zip(at, x):
                                    w2[i] = v
                                return weighted_bias(w2, cls, p, q).tolist()

                            if where == "crossing":
                                if not under_first:
                                    continue
                                marg = lambda bb: weighted_M(np.asarray(bb), factors, exact)  # noqa: E731
                                if grid32 and n >= 6:
                                    # two knobs: a weight of ordinary size brings some iteration to within ITS fp32 step of
                                    # a tie (~1e-7), the small weight at the end of the row does the rest
                                    coarse = n - min(3, n // 3) - 1
                                    g1 = place(lambda x: make([x], [coarse]), float(w[coarse]), step, marg, sign * t)
                                    x1 = float(w[coarse]) if g1 is None or not g1[0] > 0 else g1[0]
                                    got = place(lambda x: make([x1, x], [coarse, fine]), float(w[fine]), step, marg, sign * t)
                                    knobs, at = (None if got is None else [x1, got[0]]), [coarse, fine]
                                else:
                                    got = place(lambda x: make([x], [fine]), float(w[fine]), step, marg, sign * t)
                                    knobs, at = (None if got is None else [got[0]]), [fine]
                            else:
                                # probs[pick] against 1.0, margin 2 delta: first the weight of the slot itself, then (on the
                                # fp32 grid, where that knob is far too coarse) the small weight at the end of the row,
                                # which moves the average
                                slot = int(rng.integers(0, max(n - 3, 1)))
                                dl = WM.EXACT_DELTA if exact else WM.general_margins(n, WM.class_factor_spread(factors))[0]
                                got = place_pick(lambda x: make([x], [slot]), float(w[slot]), step, 2.0 * dl, sign * t, slot, n)
                                knobs, at = (None if got is None else [got[0]]), [slot]
                                if got is not None and grid32 and abs(got[1] - sign * t) > 0.04 * t:
                                    x0 = got[0]
                                    got = place_pick(lambda x: make([x0, x], [slot, fine]), float(w[fine]), step, 2.0 * dl,
                                                     sign * t, slot, n)
                                    knobs, at = (None if got is None else [x0, got[0]]), [slot, fine]
                            key0 = (where, "exact-sum" if exact else "general")
                            if got is None or not all(k > 0.0 for k in knobs) or not (0.2 * t <= abs(got[1]) <= 5.0 * t):
                                unplaced[key0] = unplaced.get(key0, 0) + 1
                                continue
                            mult, slots = got[1], got[2]
                            w2 = w.copy()
                            for i, v in zip(at, knobs):
                                w2[i] = v
                            b = weighted_bias(w2, cls, p, q)
                            ex2 = grid32 and exact_eligible(w2, b, p, q)
                            if ex2 != exact:
                                continue
                            if collect is not None:
                                collect.append(dict(n=n, w=w2, cls=cls, p=p, q=q, slots=tuple(int(x) for x in slots),
                                                    multiple=mult, exact=ex2, kind=kind, where=where))
                            alias, probs = O.alias_tables(b)
                            row = table.setdefault(key0 + (t,), [0] * 7)
                            row[0] += 1
                            check(row, b, w2, cls, factors, ex2, slots, alias, probs, n,
                                  label + f" placed at {mult:.3g} x margin ({key0[1]})")
    out("weighted: comparison, margin, multiple -> rows placed; draws on the slots of the placed comparison with r2 far "
        "from probs[pick]: declined share; all draws: declined share; wrong")
    for key in sorted(table):
        r = table[key]
        out(f"  {key[0]:8s} {key[1]:9s} x {key[2]:5.2f}: rows {r[0]:5d} placed slots {r[4]:6d} declined {r[5] / max(r[4], 1):.3f} | "
            f"all draws {r[1]:7d} declined {r[2] / max(r[1], 1):.3f} | wrong {r[3]} | fp64 loop != exact arithmetic on {r[6]} placed draws")
    out(f"  rows that could not be placed within [0.2, 5] x the multiple asked (the knob's grid is coarser than the margin): {unplaced}")
    tot = [sum(r[i] for r in table.values()) for i in range(4)]
    out(f"total {tot[1]} rows {tot[0]} declined {tot[2]} bad {tot[3]}")
    return tot


def place_pick(make_row, knob0, step_knob, M, want, slot, n, rounds=40):
    """probs[slot] - 1 = want x margin, exactly evaluated: probs = n b[slot] / sum b"""
    def g_of(k):
        b = make_row(k)
        ratios = [float(x).as_integer_ratio() for x in b]
        K = max(d for _, d in ratios)
        W = [num * (K // den) for num, den in ratios]
        T = sum(W)
        return _ratio(W[slot] * n - T, T)

    k = knob0
    # first bring the slot to the average (Newton on the exact value; the value is nearly linear in the knob)
    best = None
    for _ in range(rounds):
        g = g_of(k)
        got = g / M
        if best is None or abs(got - want) < abs(best[1] - want):
            best = (k, got, (slot,))
        if abs(got - want) <= 0.04 * abs(want):
            break
        h = abs(k) * 1e-6 + 1e-300
        k2 = step_knob(k, h)
        if k2 is None or k2 == k:
            break
        slope = (g_of(k2) - g) / (k2 - k)
        if slope == 0.0 or not math.isfinite(slope):
            break
        nk = step_knob(k, (want * M - g) / slope)
        if nk is None or nk == k or not (nk > 0.0):
            break
        k = nk
    return best


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "near"
    n_max = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    tot = (attack_near if what == "near" else attack_weighted)(n_max, per, seed)
    sys.exit(1 if tot[3] else 0)
