"""The FOLDED form of a shared-position list (include/n2v_hip.h, n2v_wedge_slots_fold; csrc/n2v_common.h ListRef and
slot_lower), as a numpy model on the CPU: a position below T is stored as it is, one from T on minus T, in the same
order; `nlow` entries lie below T.  Checked: entry k of the list is stored[k] + (k >= nlow) * T; both parts ascend; the
lower bound of any position is ONE search in one part (what list_lower_bound(ListRef) does); and the search that
enters through the eight pivots of a wedge slot (slot_lower) lands on the same place.  The lists are those of
`x in src_nbs_id` (randomwalk.py:226) for the rows of 65 536 .. 100 000 entries that the reference's own trim cap
(constants.py:6) leaves."""
import numpy as np


def fold(pos, T):
    pos = np.asarray(pos, dtype=np.int64)
    return np.where(pos < T, pos, pos - T).astype(np.uint16), int((pos < T).sum())


def folded_lower_bound(stored, nlow, T, pos, lo=0, hi=None):
    """list_lower_bound(const ListRef&, cnt, pos) / the clamped range of slot_lower"""
    cnt = len(stored) if hi is None else hi
    if nlow < len(stored):
        if pos < T:
            cnt = min(cnt, nlow)
            lo = min(lo, cnt)
            return lo + int(np.searchsorted(stored[lo:cnt], pos, side="left"))
        lo = max(lo, nlow)
        cnt = max(cnt, lo)
        return lo + int(np.searchsorted(stored[lo:cnt], pos - T, side="left"))
    return lo + int(np.searchsorted(stored[lo:cnt], pos, side="left"))


def slot_lower_model(stored, nlow, T, pick):
    """slot_lower for a list of more than 14 entries: eight pivots list[((k + 1) n) / 9], then the ninth"""
    n = len(stored)
    at = [((k + 1) * n) // 9 for k in range(8)]
    logical = [int(stored[a]) + (T if a >= nlow else 0) for a in at]
    j = sum(1 for v in logical if v < pick)
    lo = 0 if j == 0 else (j * n) // 9 + 1
    hi = n if j == 8 else ((j + 1) * n) // 9
    return folded_lower_bound(stored, nlow, T, pick, lo, hi)


def test_folded_lists_answer_every_lower_bound_with_one_search():
    rng = np.random.default_rng(3)
    cases = 0
    for T, n_max in ((65536, 100_000), (65536, 131_072), (40, 400), (2, 50), (700, 66_000)):
        for _ in range(40):
            n = int(rng.integers(T, n_max + 1)) if n_max > T else T
            n = max(min(n, T + 65536), 20)
            m = int(rng.integers(15, min(n, 4000) + 1))
            density = rng.choice(["uniform", "low", "high"])
            if density == "uniform":
                pos = np.sort(rng.choice(n, m, replace=False))
            elif density == "low":  # the shared neighbours of two hubs are hubs: low ids, low positions
                pos = np.sort(rng.choice(min(n, max(m, T // 2 + 1)), m, replace=False))
            else:
                base = max(0, n - max(m, (n - T) // 2 + 1))
                pos = base + np.sort(rng.choice(n - base, m, replace=False))
            stored, nlow = fold(pos, T)
            k = np.arange(m)
            assert np.array_equal(stored.astype(np.int64) + (k >= nlow) * T, pos)
            assert (np.diff(stored[:nlow].astype(np.int64)) > 0).all() and (np.diff(stored[nlow:].astype(np.int64)) > 0).all()
            probes = np.concatenate([pos[::7], pos[::11] + 1, rng.integers(0, n + 1, 60), [0, T - 1, T, T + 1, n - 1, n]])
            for x in probes:
                x = int(min(max(x, 0), n))
                want = int(np.searchsorted(pos, x, side="left"))
                assert folded_lower_bound(stored, nlow, T, x) == want, (T, n, m, x)
                assert slot_lower_model(stored, nlow, T, x) == want, (T, n, m, x)
            cases += 1
    assert cases == 200


def test_a_short_folded_list_packs_its_counts_into_one_halfword():
    """slots of <= 14 entries: halfword [1] = below | nlow << 4 | upper << 8 (n2v_wedge_slots_fold)"""
    rng = np.random.default_rng(5)
    for _ in range(300):
        T = int(rng.choice([2, 24, 65536]))
        n = T + int(rng.integers(0, min(65536, 3 * T) + 1))
        m = int(rng.integers(0, min(14, n) + 1))
        pos = np.sort(rng.choice(n, m, replace=False))
        rpos = int(rng.integers(0, n))
        stored, nlow = fold(pos, T)
        below, upper = int((pos < rpos).sum()), int(rpos >= T)
        hw1 = below | nlow << 4 | upper << 8
        assert hw1 < 1 << 16
        assert (hw1 & 0xf, (hw1 >> 4) & 0xf, (hw1 >> 8) & 1) == (below, nlow, upper)
        r_f = rpos if rpos < T else rpos - T
        assert r_f < 1 << 16 and r_f + upper * T == rpos


def unlisted_from_top(n, listed, T):
    """csrc/n2v_unit_core.h unlisted_from_top: the T-th slot from the top (position n - 1 first) that is not listed:
    h(k) = listed[k] - k never decreases; with i = #{k : h(k) <= c}, c = n - T - len(listed), the slot is c + i"""
    c = n - T - len(listed)
    h = np.asarray(listed, dtype=np.int64) - np.arange(len(listed))
    assert (np.diff(h) >= 0).all()
    return c + int(np.searchsorted(h, c, side="right"))


def other_from_top(n, n_ret, rpos, listed, t):
    pos = unlisted_from_top(n, listed, t)
    if n_ret > 0 and pos < rpos + n_ret:
        pos = unlisted_from_top(n, listed, t + n_ret)
    return pos


def test_the_slot_of_a_rank_is_one_search_of_the_list():
    """The alias of a pairing with "other" overfull is "the t-th `other` slot from the top" (the stack of :179-181 is
    popped from the highest position).  The kernels found it by iterating pos = n - t - (listed and return slots
    >= pos) to its fixed point (rounds 2 - 5) and now by one search (unlisted_from_top / other_from_top): both against
    a plain count over the row, on rows whose listed slots crowd the low positions as a hub's do."""
    rng = np.random.default_rng(11)
    for _ in range(300):
        n = int(rng.integers(3, 400))
        n_ret = int(rng.integers(0, 3))
        rpos = int(rng.integers(0, n - n_ret + 1)) if n_ret else 0
        free = np.array([x for x in range(n) if not (rpos <= x < rpos + n_ret)])
        m = int(rng.integers(0, max(1, len(free) - 1)))
        if rng.random() < 0.5 and m:  # crowded at the low positions
            k = min(len(free), max(m + m // 3, 1))
            listed = np.sort(free[:k][rng.permutation(k)[:m]])
        else:
            listed = np.sort(rng.choice(free, m, replace=False)) if m else np.array([], dtype=np.int64)
        special = set(listed.tolist()) | set(range(rpos, rpos + n_ret))
        others = [x for x in range(n - 1, -1, -1) if x not in special]
        unlisted = [x for x in range(n - 1, -1, -1) if x not in set(listed.tolist())]
        for t in range(1, len(unlisted) + 1):
            assert unlisted_from_top(n, listed, t) == unlisted[t - 1], (n, listed, t)
        for t in range(1, len(others) + 1):
            assert other_from_top(n, n_ret, rpos, listed, t) == others[t - 1], (n, n_ret, rpos, listed, t)
            # the fixed-point iteration of rounds 2 - 5 lands on the same slot (when it converges at all)
            c = 0
            for _it in range(len(special) + 2):
                pos = n - t - c
                c2 = sum(1 for x in special if x >= pos)
                if c2 == c:
                    break
                c = c2
            assert n - t - c == others[t - 1]
