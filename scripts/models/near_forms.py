"""Python model of the closed forms WITH MARGINS for class values that are not dyadic
(node2vec_amd/csrc/n2v_unit_near.h: near_step, lane_case_{a,b,a2,b2,a3}_near), checked against the pairing
loop of generate_alias_tables (reference randomwalk.py:157-190, restated in ref_tables: Python floats,
left-to-right sum) on random rows of the three class values: `python near_forms.py` (short rows),
`python near_forms.py big` (long rows); N2V_MODEL_TRIALS overrides the number of rows.  Every slot of every
row is asked for with a random r2.  `python near_forms.py adversarial`: EVERY (return run, shared subset)
composition of the rows of n <= N2V_MODEL_NMAX (default 6) slots x twelve (p, q), then random rows of up to
120 slots, every slot, with r2 on the nine u / 2^32 grid points around the reference's own probs[pick].
`python near_forms.py exactavg`: the second stage (near_listed_exact: the forms on the reference's own
values after the exact row sum, margin linear in n) on rows of 256 .. 20 000 slots.  Prints how many draws the forms decided, how many they left to the replay
("ambiguous") and the mismatch count (must be 0)."""
import math
import os
import random
import sys
from fractions import Fraction


def ref_tables(w):
    n = len(w); alias = [0] * n; avg = sum(w) / n; probs = [x / avg for x in w]
    under = [i for i in range(n) if probs[i] < 1.0]; over = [i for i in range(n) if not probs[i] < 1.0]
    while under and over:
        u, o = under.pop(), over.pop(); alias[u] = o; probs[o] = probs[o] + probs[u] - 1.0
        (under if probs[o] < 1.0 else over).append(o)
    return alias, probs


def fma_rem(q, step, T):  # fma(-q, step, T): the exact remainder, rounded once
    return float(Fraction(T) - Fraction(q) * Fraction(step))


def ceil_div(T, step):  # near_ceil_div
    q = math.floor(T / step); r = fma_rem(q, step, T)
    if r < 0.0: q -= 1.0; r += step
    if r >= step: q += 1.0; r -= step
    return q + 1.0 if r > 0.0 else float(q)


def floor_div(T, step):  # near_floor_div
    q = float(math.floor(T / step)); r = fma_rem(q, step, T)
    if r < 0.0: q -= 1.0
    if r >= step: q += 1.0
    return q


class Row:
    def __init__(s, n, cls):
        s.n = n; s.lst = [i for i in range(n) if cls[i] == 'M']; R = [i for i in range(n) if cls[i] == 'R']
        s.nR, s.nM = len(R), len(s.lst); s.nO = n - s.nR - s.nM; s.rpos = R[0] if R else 0
    def lower(s, pos): return sum(1 for x in s.lst if x < pos)


def case_a(G, pick, r2, V, mg, pickR, pickM, lo_pick):
    n, nR, nM, rpos, lst = G.n, G.nR, G.nM, G.rpos, G.lst
    EM, ER, D = V['M'] - 1.0, V['R'] - 1.0, 1.0 - V['O']
    if not D > mg or (nM > 0 and not EM > mg) or (nR > 0 and not ER > mg): return None
    mA = nM - G.lower(rpos) if (nR > 0 and nM > 0) else nM
    N = nM + nR
    def X_of(i):
        if i <= mA: return i * EM
        if i <= mA + nR: return mA * EM + (i - mA) * ER
        return mA * EM + nR * ER + (i - mA - nR) * EM
    def pos_of(i):
        if i <= mA: return lst[nM - i]
        if i <= mA + nR: return rpos + nR - (i - mA)
        return lst[nM - (i - nR)]
    if not pickR and not pickM:
        above_r = max(0, min(nR, rpos + nR - 1 - pick))
        r = float((n - 1 - pick) - (nM - lo_pick) - above_r); T = r * D
        if not r >= 1.0: return pos_of(1)
        if mA > 0 and mA * EM >= T: i = ceil_div(T, EM)
        else:
            X1 = mA * EM
            if nR > 0 and X1 + nR * ER >= T: i = mA + ceil_div(T - X1, ER)
            elif nM > 0: i = mA + nR + ceil_div(T - X1 - nR * ER, EM)
            else: return None
        if not i >= 1.0 or i > N: return None
        if not X_of(i) - T > mg: return None
        if i > 1.0 and not T - X_of(i - 1.0) > mg: return None
        return pos_of(int(i))
    if pickR: i0 = mA + (nR - (pick - rpos))
    else:
        d = nM - lo_pick; i0 = d if d <= mA else d + nR
    if i0 < 1 or i0 > N: return None
    if i0 == N: return pick if r2 < 1.0 - mg - 1e-9 else None
    X = X_of(float(i0)); xq = floor_div(X, D); rem0 = fma_rem(xq, D, X)
    if not rem0 > mg or not D - rem0 > mg: return None
    prob = 1.0 + (rem0 - D)
    if abs(prob - r2) < 1e-9 + mg: return None
    return pick if r2 < prob else pos_of(i0 + 1)


def case_b(G, pick, r2, V, mg, pickR, pickM, lo_pick):
    n, nR, nM, nO, rpos, lst = G.n, G.nR, G.nM, G.nO, G.rpos, G.lst
    e, dR, dM = V['O'] - 1.0, 1.0 - V['R'], 1.0 - V['M']
    if nO <= 0 or not e > mg or (nM > 0 and not dM > mg) or (nR > 0 and not dR > mg): return None
    mA = nM - G.lower(rpos) if (nR > 0 and nM > 0) else nM
    S = nM + nR
    def Y_of(j):
        if j <= mA: return j * dM
        if j <= mA + nR: return mA * dM + (j - mA) * dR
        return mA * dM + nR * dR + (j - mA - nR) * dM
    def specials_ge(pos): return (nM - G.lower(pos)) + max(0, min(nR, rpos + nR - pos))
    def other_pos(t):
        c = 0
        for _ in range(64):
            c2 = specials_ge(n - t - c)
            if c2 == c: return n - t - c
            c = c2
        return None
    if pickR or pickM:
        if pickR: j = mA + (nR - (pick - rpos))
        else:
            d = nM - lo_pick; j = d if d <= mA else d + nR
        if j < 1 or j > S: return None
        t = 1.0
        if j > 1:
            Yp = Y_of(float(j - 1)); t = ceil_div(Yp, e)
            if not t * e - Yp > mg: return None
            if t > 1.0 and not Yp - (t - 1.0) * e > mg: return None
        if not t >= 1.0 or t > nO: return None
        return other_pos(int(t))
    ar = max(0, min(nR, rpos + nR - 1 - pick))
    t = (n - pick) - (nM - lo_pick) - ar
    if t < 1 or t > nO: return None
    if t == nO: return pick if r2 < 1.0 - mg - 1e-9 else None
    T = float(t) * e
    if mA > 0 and mA * dM > T: j = floor_div(T, dM) + 1.0
    else:
        Y1 = mA * dM
        if nR > 0 and Y1 + nR * dR > T: j = mA + floor_div(T - Y1, dR) + 1.0
        elif nM > 0 and T - Y1 - nR * dR >= 0.0: j = mA + nR + floor_div(T - Y1 - nR * dR, dM) + 1.0
        else: return None
    if not j >= 1.0 or j > S: return None
    if not Y_of(j) - T > mg: return None
    if j > 1.0 and not T - Y_of(j - 1.0) > mg: return None
    prob = 1.0 + (T - Y_of(j))
    if abs(prob - r2) < 1e-9 + mg: return None
    if r2 < prob: return pick
    return other_pos(t + 1)


class Two:  # TwoOnStack
    def __init__(s, G):
        s.G = G; mA = G.nM - G.lower(G.rpos)
        s.rho = (G.n - G.rpos - G.nR) - mA; s.nS = G.nO + G.nR
    def stack_pos(s, t):
        G = s.G
        if t < 1 or t > s.nS: return 0
        c = 0
        for _ in range(64):
            c2 = G.nM - G.lower(G.n - t - c)
            if c2 == c: return G.n - t - c
            c = c2
        seen = 0
        for i in range(G.n - 1, -1, -1):
            if i not in G.lst:
                seen += 1
                if seen == t: return i
        return 0
    def stack_rank(s, pick, pickR, lo_pick):
        G = s.G
        if pickR: return s.rho + (G.rpos + G.nR - pick)
        return (G.n - pick) - (G.nM - lo_pick)


def case_a2(G, pick, r2, V, mg, pickR, pickM, lo_pick):
    nR, nM, lst = G.nR, G.nM, G.lst
    EM, D, DR = V['M'] - 1.0, 1.0 - V['O'], 1.0 - V['R']
    if nR <= 0 or nM <= 0 or not D > mg or not DR > mg or not EM > mg: return None
    T2 = Two(G); rho = T2.rho
    def Def(k):
        if k <= rho: return k * D
        if k <= rho + nR: return rho * D + (k - rho) * DR
        return rho * D + nR * DR + (k - rho - nR) * D
    if not pickM:
        k = T2.stack_rank(pick, pickR, lo_pick) - 1
        if k < 0 or k >= T2.nS: return None
        T = Def(float(k)); i = 1.0
        if k > 0:
            i = ceil_div(T, EM)
            if not i * EM - T > mg: return None
            if i > 1.0 and not T - (i - 1.0) * EM > mg: return None
        if not i >= 1.0 or i > nM: return None
        return lst[nM - int(i)]
    i0 = nM - lo_pick
    if i0 < 1 or i0 > nM: return None
    if i0 == nM: return pick if r2 < 1.0 - mg - 1e-9 else None
    X = float(i0) * EM
    if rho * D > X: k = floor_div(X, D) + 1.0
    else:
        Y1 = rho * D
        if Y1 + nR * DR > X: k = rho + floor_div(X - Y1, DR) + 1.0
        elif X - Y1 - nR * DR >= 0.0: k = rho + nR + floor_div(X - Y1 - nR * DR, D) + 1.0
        else: return None
    if not k >= 1.0 or k > T2.nS: return None
    if not Def(k) - X > mg: return None
    if not X - Def(k - 1.0) > mg: return None
    prob = 1.0 + (X - Def(k))
    if abs(prob - r2) < 1e-9 + mg: return None
    return pick if r2 < prob else lst[nM - (i0 + 1)]


def case_b2(G, pick, r2, V, mg, pickR, pickM, lo_pick):
    nR, nM, nO = G.nR, G.nM, G.nO
    e, eR, dM = V['O'] - 1.0, V['R'] - 1.0, 1.0 - V['M']
    if nR <= 0 or nM <= 0 or nO <= 0 or not e > mg or not eR > mg or not dM > mg: return None
    T2 = Two(G); rho = T2.rho
    def Xo(t):
        if t <= rho: return t * e
        if t <= rho + nR: return rho * e + (t - rho) * eR
        return rho * e + nR * eR + (t - rho - nR) * e
    if pickM:
        j = nM - lo_pick
        if j < 1 or j > nM: return None
        t = 1.0
        if j > 1:
            Yp = float(j - 1) * dM
            if rho * e >= Yp: t = ceil_div(Yp, e)
            else:
                X1 = rho * e
                if X1 + nR * eR >= Yp: t = rho + ceil_div(Yp - X1, eR)
                else: t = rho + nR + ceil_div(Yp - X1 - nR * eR, e)
            if not t >= 1.0 or t > T2.nS: return None
            if not Xo(t) - Yp > mg: return None
            if t > 1.0 and not Yp - Xo(t - 1.0) > mg: return None
        if not t >= 1.0 or t > T2.nS: return None
        return T2.stack_pos(int(t))
    t = T2.stack_rank(pick, pickR, lo_pick)
    if t < 1 or t > T2.nS: return None
    if t == T2.nS: return pick if r2 < 1.0 - mg - 1e-9 else None
    T = Xo(float(t)); j = floor_div(T, dM) + 1.0
    if not j >= 1.0 or j > nM: return None
    if not j * dM - T > mg: return None
    if not T - (j - 1.0) * dM > mg: return None
    prob = 1.0 + (T - j * dM)
    if abs(prob - r2) < 1e-9 + mg: return None
    if r2 < prob: return pick
    return T2.stack_pos(t + 1)


def case_a3(G, pick, r2, V, mg, pickR, pickM, lo_pick):
    n, nR, nM, rpos = G.n, G.nR, G.nM, G.rpos
    ER, D, DM = V['R'] - 1.0, 1.0 - V['O'], 1.0 - V['M']
    if nR <= 0 or nM <= 0 or not ER > mg or not D > mg or not DM > mg: return None
    if pickR: return pick if (rpos + nR - pick == nR and r2 < 1.0 - mg - 1e-9) else None
    m_above = nM - lo_pick - (1 if pickM else 0)
    ar = max(0, min(nR, rpos + nR - 1 - pick))
    o_above = (n - 1 - pick) - ar - m_above
    T = float(o_above) * D + float(m_above) * DM
    iq = floor_div(T, ER)
    if o_above + m_above > 0:
        rem = fma_rem(iq, ER, T)
        if not rem > mg or not ER - rem > mg: return None
    if iq >= nR: return 0
    return rpos + nR - 1 - int(iq)


def near_step(n, cls, pick, r2, b):
    """near_step of n2v_unit_near.h: the slot, or None (the exact row sum and the replays decide)"""
    G = Row(n, cls); nR, nM, nO = G.nR, G.nM, G.nO
    if (nR == n or nM == n or nO == n) and n < (1 << 20): return pick
    pickR, pickM = cls[pick] == 'R', cls[pick] == 'M'
    b_pick = b[cls[pick]]
    approx = (float(nR) * b['R'] + float(nM) * b['M'] + float(nO) * b['O']) / float(n)
    eps = (float(n) + 8.0) * 4.5e-16
    lo_f, hi_f = 1.0 - 2.0 * eps, 1.0 + 2.0 * eps
    dec = lambda c, k: k == 0 or b[c] < approx * lo_f or b[c] > approx * hi_f
    pp = b_pick / approx
    not_accepted = b_pick > approx * hi_f or (b_pick < approx * lo_f and r2 > pp * hi_f)
    if not (dec('R', nR) and dec('M', nM) and dec('O', nO) and not_accepted): return None
    uR, uM, uO = b['R'] < approx, b['M'] < approx, b['O'] < approx
    any_under = (nR and uR) or (nM and uM) or (nO and uO)
    any_over = (nR and not uR) or (nM and not uM) or (nO and not uO)
    if not any_under or not any_over: return None
    arr = 0
    if uO and not (nR and uR) and not (nM and uM): arr = 1
    elif not uO and nO > 0 and (not nR or uR) and (not nM or uM): arr = 2
    elif uO and nR and uR and nM and not uM: arr = 3
    elif not uO and nO > 0 and nR and not uR and nM and uM: arr = 4
    elif uO and nR and not uR and nM and uM: arr = 5
    V = {c: b[c] / approx for c in 'RMO'}
    vmax = max([V['O']] * (nO > 0) + [V['R']] * (nR > 0) + [V['M']] * (nM > 0))
    mg = 5e-15 * float(n) * (float(n) + 8.0) * vmax
    lo_pick = G.lower(pick)
    f = {1: case_a, 2: case_b, 3: case_a2, 4: case_b2, 5: case_a3}.get(arr)
    return None if f is None else f(G, pick, r2, V, mg, pickR, pickM, lo_pick)


def near_step_exact(n, cls, pick, r2, b):
    """near_listed_exact of n2v_unit_near.h: the forms on the reference's OWN values (avg = the
    left-to-right sum / n, v = b / avg) with the margin that is linear in n; None = replay"""
    G = Row(n, cls); nR, nM, nO = G.nR, G.nM, G.nO
    w = [b[c] for c in cls]
    avg = sum(w) / n
    uR, uM, uO = b['R'] < avg, b['M'] < avg, b['O'] < avg
    any_under = (nR and uR) or (nM and uM) or (nO and uO)
    any_over = (nR and not uR) or (nM and not uM) or (nO and not uO)
    if not any_under or not any_over: return None
    arr = 0
    if uO and not (nR and uR) and not (nM and uM): arr = 1
    elif not uO and nO > 0 and (not nR or uR) and (not nM or uM): arr = 2
    elif uO and nR and uR and nM and not uM: arr = 3
    elif not uO and nO > 0 and nR and not uR and nM and uM: arr = 4
    elif uO and nR and not uR and nM and uM: arr = 5
    V = {c: b[c] / avg for c in 'RMO'}
    vmax = max([V['O']] * (nO > 0) + [V['R']] * (nR > 0) + [V['M']] * (nM > 0))
    mg = 2e-14 * float(n) * (vmax + 1.0)
    f = {1: case_a, 2: case_b, 3: case_a2, 4: case_b2, 5: case_a3}.get(arr)
    return None if f is None else f(G, pick, r2, V, mg, cls[pick] == 'R', cls[pick] == 'M', G.lower(pick))


STEP = [None]  # the step function under test (near_step; near_step_exact in mode "exactavg")


def check_row(n, cls, b, picks, adversarial, stats):
    """every slot of `picks` of one row against the reference's loop.  adversarial: r2 on the nine
    u / 2^32 grid points around the reference's own probs[pick] -- the draws a random r2 never
    hits, where the decision  r2 < probs[pick]  is as close as the uniform stream allows"""
    w = [b[c] for c in cls]
    alias, probs = ref_tables(w)
    avg = sum(w) / n
    for pick in picks:
        if adversarial:
            u0 = int(math.floor(probs[pick] * 2.0 ** 32))
            r2s = [min(max(u0 + d, 0), 2 ** 32 - 1) / 2 ** 32 for d in range(-4, 5)]
        else:
            r2s = [random.getrandbits(32) / 2 ** 32]
        for r2 in r2s:
            p0 = w[pick] / avg
            if p0 < 1.0 and r2 < p0: continue  # the quick accept (the kernel's own margin test comes first)
            want = pick if r2 < probs[pick] else alias[pick]
            got = (STEP[0] or near_step)(n, cls, pick, r2, b)
            stats[0] += 1
            if got is None: stats[1] += 1
            elif got != want:
                stats[2] += 1
                if stats[2] < 6: print("MISMATCH", n, b, ''.join(cls) if n < 100 else "", pick, r2, want, got)


def dyadic_pq(p, q):
    return all((1.0 / x) == 2.0 ** round(math.log2(1.0 / x)) for x in (p, q))


def random_row(n, p, q):
    cls = ['O'] * n
    nR = random.choice([0, 1, 1, 1, 2, 3]) if n > 3 else random.choice([0, 1])
    rp = random.randint(0, n - nR)
    for k in range(nR): cls[rp + k] = 'R'
    dens = random.choice([0.0, 0.02, 0.1, 0.3, 0.5, 0.9])
    for i in range(n):
        if cls[i] == 'O' and random.random() < dens: cls[i] = 'M'
    if q == 1.0: cls = [c if c != 'M' else 'O' for c in cls]  # need_mem false: no listed slots
    if p == q: cls = [c if c != 'R' else 'O' for c in cls]    # merge_r: the return slot IS an "other" slot
    return cls


def compositions(n):
    """every row of n slots: a return run (0 .. n consecutive slots anywhere) and any subset of the
    remaining slots shared"""
    runs = [(0, 0)] + [(rp, k) for k in range(1, n + 1) for rp in range(0, n - k + 1)]
    for rp, k in runs:
        rest = [i for i in range(n) if not (rp <= i < rp + k)]
        for mask in range(1 << len(rest)):
            cls = ['O'] * n
            for i in range(rp, rp + k): cls[i] = 'R'
            for j, i in enumerate(rest):
                if mask >> j & 1: cls[i] = 'M'
            yield cls


if __name__ == "__main__":  # (importable: scripts/models/margin_adversary.py uses the forms above)
    random.seed(int(os.environ.get("N2V_MODEL_SEED", 7)))
    mode = sys.argv[1] if len(sys.argv) > 1 else "short"
    big = mode == "big"
    VALS = [0.7, 1.3, 3.0, 0.3, 1.5, 6.0, 0.75, 1.0 / 3.0, 2.0 / 3.0, 5.0, 0.2, 0.6, 1.2, 2.5, 7.0, 1.0 / 7.0, 10.0, 0.1, 37.5, 1.0]
    stats = [0, 0, 0]  # draws, left to the replay, wrong
    rows = 0
    if mode == "adversarial":
        #  every (return run, shared subset) composition of the rows of n <= NMAX slots x
        # twelve (p, q) that are not dyadic, then random rows of up to 120 slots, every slot, adversarial r2
        PQ = [(0.7, 3.0), (3.0, 0.7), (1.3, 1.3), (0.3, 0.7), (3.0, 1.0), (1.0 / 3.0, 2.0 / 3.0), (1.5, 6.0), (6.0, 1.5),
              (0.75, 1.2), (5.0, 0.2), (7.0, 1.0 / 7.0), (37.5, 0.6)]
        nmax = int(os.environ.get("N2V_MODEL_NMAX", 6))
        for n in range(1, nmax + 1):
            for cls0 in compositions(n):
                for p, q in PQ:
                    cls = list(cls0)
                    if q == 1.0: cls = [c if c != 'M' else 'O' for c in cls]
                    check_row(n, cls, {'R': 1.0 / p, 'M': 1.0, 'O': 1.0 / q}, range(n), True, stats)
                    rows += 1
        trials = int(os.environ.get("N2V_MODEL_TRIALS", 600))
        k = 0
        while k < trials:
            n = random.randint(1, 120)
            p, q = random.choice(VALS), random.choice(VALS)
            if dyadic_pq(p, q): continue
            check_row(n, random_row(n, p, q), {'R': 1.0 / p, 'M': 1.0, 'O': 1.0 / q}, range(n), True, stats)
            k += 1; rows += 1
    elif mode == "exactavg":
        # the second stage (round 5): rows of 256 .. N2V_MODEL_NMAX slots (default 20 000), the forms on the
        # reference's own values with the LINEAR margin, r2 random and adversarial, 48 slots per row + every
        # return / shared slot of the row's first 200
        STEP[0] = near_step_exact
        trials = int(os.environ.get("N2V_MODEL_TRIALS", 60))
        nmax = int(os.environ.get("N2V_MODEL_NMAX", 20000))
        k = 0
        while k < trials:
            n = int(math.exp(random.uniform(math.log(256), math.log(nmax))))
            p, q = random.choice(VALS), random.choice(VALS)
            if dyadic_pq(p, q): continue
            cls = random_row(n, p, q)
            if random.random() < 0.5:  # a sparse list, as the rows of a hub have
                cls = [c if c != 'M' or random.random() < 0.05 else 'O' for c in cls]
            special = [i for i in range(n) if cls[i] != 'O'][:200]
            picks = sorted(set(random.sample(range(n), 48) + special))
            bb = {'R': 1.0 / p, 'M': 1.0, 'O': 1.0 / q}
            check_row(n, cls, bb, picks, True, stats)
            rnd = [0, 0, 0]
            check_row(n, cls, bb, picks, False, rnd)
            STEP[0] = near_step  # the first stage on the same draws, for comparison
            first = [0, 0, 0]
            check_row(n, cls, bb, picks, False, first)
            STEP[0] = near_step_exact
            for i in range(3): stats[i] += rnd[i]
            random_draws = [a + b for a, b in zip(globals().get("random_draws", [0, 0, 0, 0]), rnd + [first[1]])]
            k += 1; rows += 1
        print("random r2 only: draws", random_draws[0], "left to the replay by the second stage", random_draws[1],
              "(by the first stage, counts + n^2 margin:", random_draws[3], ") bad", random_draws[2])
    else:
        trials = int(os.environ.get("N2V_MODEL_TRIALS", 3000 if not big else 150))
        while rows < trials:
            n = random.randint(200, 2500) if big else random.randint(1, 70)
            p, q = random.choice(VALS), random.choice(VALS)
            if dyadic_pq(p, q): continue  # (dyadic: the other forms)
            cls = random_row(n, p, q)
            check_row(n, cls, {'R': 1.0 / p, 'M': 1.0, 'O': 1.0 / q},
                      range(n) if not big else random.sample(range(n), 60), False, stats)
            rows += 1
    print("rows", rows)
    print("total", stats[0], "ambiguous", stats[1], "bad", stats[2])
    sys.exit(1 if stats[2] else 0)
