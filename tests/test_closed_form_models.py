"""The closed forms of csrc/n2v_unit_core.h were derived and first checked as Python models
(scripts/models/): exact integer bucket arithmetic against the pairing loop of
generate_alias_tables (reference randomwalk.py:175-189) on random rows of the three class values.
The HIP code itself is compared with the oracle in the -m gpu tests; this keeps the models (the
derivation DESIGN.md cites) running: no mismatch on a few thousand random rows each."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = ["case_a_jump.py", "case_b_jump.py", "case_b_replay.py", "case_a2.py", "case_b2.py", "case_a3.py",
          "near_forms.py"]  # (near_forms: the closed forms with margins for values that are not dyadic)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("rows", ["short", "long"])
def test_model_agrees_with_the_reference_loop(name, rows):
    if rows == "long" and name == "case_a_jump.py":
        pytest.skip("this model draws short rows only")
    env = dict(os.environ, N2V_MODEL_TRIALS="4000" if rows == "short" else ("300" if name != "near_forms.py" else "150"))
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "models", name)] + (["big"] if rows == "long" else [])
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    last = res.stdout.strip().splitlines()[-1]
    assert last.startswith("total") and " bad" in last
    assert int(last.split()[1]) > 0  # rows were really drawn


def test_closed_forms_with_margins_on_adversarial_draws():
    """near_forms.py adversarial: every (return run, shared subset) composition of
    the rows of n <= 6 slots x twelve (p, q) that are not dyadic + random rows of up to 120 slots, every
    slot, r2 on the nine u / 2^32 grid points around the reference's own probs[pick] -- the draws where
    r2 < probs[pick] is as close as the uniform stream allows.  The forms may decline; they must never
    be wrong.  (n <= 7 and 3 000 rows: profiles/r7_near_model_adversarial.log, 1.96 M draws.)"""
    env = dict(os.environ, N2V_MODEL_NMAX="6", N2V_MODEL_TRIALS="600")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "models", "near_forms.py"), "adversarial"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    last = res.stdout.strip().splitlines()[-1].split()
    assert last[0] == "total" and last[-2:] == ["bad", "0"]
    assert int(last[1]) > 400_000 and int(last[3]) < int(last[1])  # many draws, not all declined


def test_closed_forms_on_the_exact_average_of_long_rows():
    """near_forms.py exactavg (round 5): the second stage of the closed forms with margins -- after the row has
    been added up in the reference's order the forms run on the reference's OWN values with a margin that is
    linear in n (near_listed_exact) -- on rows of 256 .. 20 000 slots, random and adversarial r2: never wrong."""
    env = dict(os.environ, N2V_MODEL_TRIALS="24", N2V_MODEL_NMAX="20000")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "models", "near_forms.py"), "exactavg"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    last = res.stdout.strip().splitlines()[-1].split()
    assert last[0] == "total" and last[-2:] == ["bad", "0"] and int(last[1]) > 5000


def test_weighted_rows_decided_with_margins_model():
    """weighted_margins.py (round 5): the model of walk_weighted_margin_kernel -- the one slot a draw asks of a
    WEIGHTED row's table decided from sums over the row (the k-th overfull slot is demoted where the running sum
    of the deficits passes that of the excesses) with margins, the row sum taken from the stored row sum and
    the shared / return slots -- against the table generate_alias_tables builds: random, heavy-tailed, few-valued
    and nearly tied rows, uniforms drawn, exactly at the table's threshold and 1e-6 .. 1e-14 beside it.  Every
    draw is either the table's or left undecided."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "models", "weighted_margins.py"), "400", "3"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert " 0 wrong" in res.stdout and "WRONG" not in res.stdout, res.stdout


def test_a_class_exactly_on_the_average_model():
    """lane_case_b2_jump on rows whose "other" slots have excess 0 (scripts/models/flat_b2.py)"""
    env = dict(os.environ, N2V_MODEL_TRIALS="4000")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "models", "flat_b2.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    last = res.stdout.strip().splitlines()[-1]
    assert last.startswith("total") and last.endswith("bad 0") and int(last.split()[1]) > 10000


def test_layered_sampler_model_gives_every_slot_its_weight():
    """the layer decomposition of fast mode's sampler (csrc/n2v_walk_fast.hip, kClassFirst) in exact
    rational arithmetic: P(slot) == weight / sum of weights for every ordering of 1/p, 1, 1/q"""
    env = dict(os.environ, N2V_MODEL_TRIALS="3000")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "models", "layered_sampler.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    last = res.stdout.strip().splitlines()[-1]
    assert last.startswith("total") and last.endswith(" 0 bad") and int(last.split()[1]) == 3000


def _adversary(what, n_max, per, seed):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "models", "margin_adversary.py"), what,
                          str(n_max), str(per), str(seed)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "WRONG" not in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    last = res.stdout.strip().splitlines()[-1].split()
    assert last[0] == "total" and last[-2:] == ["bad", "0"], res.stdout[-500:]
    return res.stdout, int(last[1]), int(last[3])


def test_closed_forms_with_margins_on_rows_placed_at_the_margin_by_exact_arithmetic():
    """margin_adversary.py near (round 6): the reference's loop run in exact rational arithmetic
    on a row, q moved until the iteration nearest to a tie -- X_i - k D of DESIGN.md 5 -- sits at +-{0.01, 0.05, 0.25,
    0.5, 0.9, 1.1, 2, 10} x the margin of the counts stage (5e-15 n (n + 8) vmax) or of the exact-sum stage
    (2e-14 n (vmax + 1)), rows of 8 ... 10^4 slots in all five class arrangements.  Both stages as the kernel chains
    them: a decided draw is the reference's fp64 loop's (generate_alias_tables, randomwalk.py:172-189); and outside the
    smaller margin the placed slots ARE decided (the procedure is not vacuous).  Rows of 10^5 slots:
    profiles/r10c_margin_adversary_near.log."""
    out, draws, rows = _adversary("near", 10_000, 1, 11)
    assert draws > 40_000 and rows > 400
    inside = [float(l.split("declined")[1].split()[0]) for l in out.splitlines() if "margin ~n  " in l and
              any(f"x {t:5.2f}:" in l for t in (0.25, 0.5, 0.9))]
    outside = [float(l.split("declined")[1].split()[0]) for l in out.splitlines() if "margin ~n  " in l and
               any(f"x {t:5.2f}:" in l for t in (1.1, 2.0, 10.0))]
    assert len(inside) == 3 and len(outside) == 3 and min(inside) > 0.3 and max(outside) < 0.2, (inside, outside)


def test_weighted_decision_on_rows_placed_at_the_margin_by_exact_arithmetic():
    """margin_adversary.py weighted: the stored weight of one slot moved (on the fp32 grid where the row is to keep an
    exact row sum) until the crossing E_k - D_j, or probs[pick] - 1, sits at the same multiples of the general margins
    (kfac 16 n^2 2^-52; 2 delta) or of the margins of an exact sum (linear in n); fp32, fp64, 24-decade rows, and
    integer rows whose sums tie exactly.  Long rows go through the model of the block summaries."""
    out, draws, rows = _adversary("weighted", 10_000, 1, 12)
    assert draws > 25_000 and rows > 200
    sharp = [l for l in out.splitlines() if "crossing exact-sum" in l]
    inside = [float(l.split("declined")[1].split()[0]) for l in sharp if any(f"x {t:5.2f}:" in l for t in (0.25, 0.5, 0.9))]
    outside = [float(l.split("declined")[1].split()[0]) for l in sharp if any(f"x {t:5.2f}:" in l for t in (1.1, 2.0, 10.0))]
    assert len(inside) == 3 and len(outside) == 3 and min(inside) > 0.2 and max(outside) < 0.1, (inside, outside)
