"""Pins the CPU oracle (oracle/n2v_oracle.c) to the reference.

Every expected value here was produced by the reference's own
node2vec/randomwalk.py (tests/golden/gen_golden.py) or is a known answer of
the reference's tests/test_randomwalk.py (cited per test).
"""
import numpy as np
import pytest

from conftest import load_golden


def test_rng_stream_matches_generator(oracle):
    for k in load_golden("rng_kat.json"):
        assert list(oracle.uniform_bits(k["seed"], k["key"], k["step"])) == k["u"]


def test_generate_alias_tables_golden(oracle):
    """randomwalk.py:157-190; bit-exact alias AND probs (fp64)."""
    for c in load_golden("g1_alias_tables.json"):
        alias, probs = oracle.alias_tables(c["weights"])
        assert alias == c["alias"]
        assert probs == c["probs"]  # exact fp64 equality


def test_generate_alias_tables_reference_kat(oracle):
    """tests/test_randomwalk.py:131-152"""
    for w, (alias, probs) in (
        ([0.5, 0.8, 1.0], ([2, 0, 1], [0.6521739, 1.0, 0.9565217])),
        ([0.5, 0.2], ([0, 0], [1.0, 0.5714285714285715])),
        ([0.2], ([0], [1.0])),
        ([1.0], ([0], [1.0])),
    ):
        a, p = oracle.alias_tables(w)
        assert a == alias
        np.testing.assert_almost_equal(p, probs, decimal=7)


def test_alias_tables_ulp_leftover_quirk(oracle):
    """SURVEY 7 'hard parts': [0.1]*10 keeps alias 0 and probs 1+ulp."""
    a, p = oracle.alias_tables([0.1] * 10)
    assert a == [0] * 10 and p == [1.0000000000000002] * 10


def test_alias_tables_zero_division(oracle):
    with pytest.raises(ZeroDivisionError):
        oracle.alias_tables([])
    with pytest.raises(ZeroDivisionError):
        oracle.alias_tables([0.0, 0.0])


def test_generate_edge_alias_tables_golden(oracle):
    """randomwalk.py:193-232 incl. tests/test_randomwalk.py:158-160"""
    g = load_golden("g2_edge_alias_tables.json")
    for c in g["cases"]:
        alias, probs = oracle.edge_alias_tables(c["src_id"], c["src_nbs"], c["dst_ids"],
                                                c["dst_w"], c["p"], c["q"])
        assert alias == c["alias"]
        assert probs == c["probs"]
    a, p = oracle.edge_alias_tables(3, [], [1, 3], [0.5, 1.0], 2.0, 4.0)
    assert a == [1, 0]
    np.testing.assert_almost_equal(p, [0.4, 1.0], decimal=7)


def test_generate_edge_alias_tables_errors(oracle):
    """tests/test_randomwalk.py:184-189: the three ValueError cases"""
    g = load_golden("g2_edge_alias_tables.json")
    assert len(g["errors"]) == 9
    for c in g["errors"]:
        assert c["raises"] == "ValueError"
        with pytest.raises(ValueError):
            oracle.edge_alias_tables(c["src_id"], c["src_nbs"], c["dst_ids"], c["dst_w"],
                                     c["p"], c["q"])


def test_samplers_golden(oracle):
    """randomwalk.py:70-99"""
    g = load_golden("g3_samplers.json")
    for t in g["tables"]:
        for d in t["draws"]:
            if "two" in d:
                assert oracle.sampling_from_alias(t["alias"], t["probs"], d["r1"], d["r2"]) == d["two"]
            else:
                assert oracle.sampling_from_alias_wiki(t["alias"], t["probs"], d["r1"]) == d["wiki"]
    k = g["seed20"]  # tests/test_randomwalk.py:65-72, 83-90
    assert (k["r1"], k["r2"]) == (0.9056396761745207, 0.6862541570267026)
    for c in k["cases"]:
        assert c["ids"][oracle.sampling_from_alias_wiki(c["alias"], c["probs"], k["r1"])] == c["wiki"]
        assert c["ids"][oracle.sampling_from_alias(c["alias"], c["probs"], k["r1"], k["r2"])] == c["two"]
    assert [c["two"] for c in k["cases"]] == [22, 122]


def test_path_append_golden(oracle):
    """randomwalk.py:123-153 incl. the first-step rule, tests/test_randomwalk.py:97-101"""
    for c in load_golden("g3_samplers.json")["path_append"]:
        got = oracle.path_append(c["path"], c["dst_nbs"], c["alias"], c["probs"], c["r1"], c["r2"])
        assert got == c["result"]


def test_next_step_reference_kat(oracle):
    """tests/test_randomwalk.py:268-306 (MT seeds 1000/10/20 -> dst 4, 3, 3),
    the recorded MT uniforms fed to the oracle's functions."""
    g = load_golden("g5_next_step.json")
    for c in g["next_step"]:
        if c["src"] < 0:  # randomwalk.py:320-321
            alias, probs = oracle.alias_tables(c["dst_w"])
        else:
            alias, probs = oracle.edge_alias_tables(c["src"], c["src_nbs"], c["dst_ids"],
                                                    c["dst_w"], c["p"], c["q"])
        path = oracle.path_append(c["path"], c["dst_ids"], alias, probs, c["r1"], c["r2"])
        assert path == c["out_path"]
        assert (path[-2], path[-1]) == (c["out_src"], c["out_dst"])
    assert [c["out_dst"] for c in g["next_step"]] == [4, 3, 3]
    # initiate_random_walk / to_path shapes, tests/test_randomwalk.py:245-264, 310-324
    assert [r["path"] for r in g["initiate"]] == [[-1, 3], [-2, 3], [-3, 3], [-1, 2], [-2, 2], [-3, 2]]
    assert [r["src"] for r in g["to_path"]] == [1, 1, 0]


def _run_walk_case(oracle, c, n_threads=1):
    rowptr, col, w = oracle.csr_from_edges(c["edges"])
    if c["name"].endswith("_fp64"):  # weights that are not fp32 values must reach the oracle as fp64
        assert w.dtype == np.float64
        assert w.tolist() == [e[2] for e in sorted(c["edges"], key=lambda e: (e[0], e[1]))]
    nv = len(rowptr) - 1
    start = list(range(nv)) if c["walk_seed"] is None else sorted(set(c["walk_seed"]))
    walks, valid = oracle.random_walk(rowptr, col, w, start, c["num_walks"], c["walk_length"],
                                      c["p"], c["q"], c["seed"], n_threads)
    got = {}
    for i, s in enumerate(start):
        for o in range(c["num_walks"]):
            r = i * c["num_walks"] + o
            if valid[r]:
                got[(s, o + 1)] = walks[r].tolist()
    return got


@pytest.mark.parametrize("threads", [1, 3])
def test_whole_walks_match_reference(oracle, threads):
    """fugue.py:130-155 driven through the reference's own transformers with the
    build's uniform stream replayed (G4/G7): identical walks, identical drops."""
    for c in load_golden("g4_walks.json"):
        got = _run_walk_case(oracle, c, threads)
        want = {(w["start"], w["ordinal"]): w["walk"] for w in c["walks"]}
        assert got.keys() == want.keys(), c["name"]
        for k in want:
            assert got[k] == want[k], (c["name"], k)
            assert len(got[k]) == c["walk_length"] + 1
            assert got[k][0] == k[0]  # to_path: src = path[0]


def test_fp64_weights_are_not_narrowed(oracle):
    """The decimal_weights_* fixtures were generated so that rounding the weights to fp32 changes
    walks IN THE REFERENCE (gen_golden.py records how many); the oracle fed fp32-rounded weights
    must therefore disagree with the fixture, and fed the fp64 weights must not."""
    cases = [c for c in load_golden("g4_walks.json") if "walks_differing_with_fp32_weights" in c]
    assert len(cases) == 2
    for c in cases:
        assert c["walks_differing_with_fp32_weights"] >= 8
        want = {(w["start"], w["ordinal"]): w["walk"] for w in c["walks"]}
        assert _run_walk_case(oracle, c) == want
        narrowed = dict(c, name="narrowed",
                        edges=[(a, b, float(np.float32(w))) for a, b, w in c["edges"]])
        got32 = _run_walk_case(oracle, narrowed)
        assert got32.keys() == want.keys()
        assert sum(got32[k] != want[k] for k in want) == c["walks_differing_with_fp32_weights"]


def test_sink_semantics(oracle):
    """G7: graph 0->1, 1->2, 3->0, W=1, L=2 keeps only [0,1,2] and [3,0,1]."""
    c = load_golden("g4_walks.json")[0]
    assert c["name"] == "sink3"
    got = _run_walk_case(oracle, c)
    assert sorted(got.values()) == [[0, 1, 2], [3, 0, 1]]


def test_transition_probs_match_reference_tables(oracle):
    """G6: the oracle's pi(x|s,v) equals the distribution implied by the
    reference's alias tables (to rounding)."""
    g = load_golden("g2_edge_alias_tables.json")
    n = 0
    for c in g["cases"]:
        if "implied" not in c:
            continue
        # one-row graph for v plus a row for s
        rowptr, col, w = oracle.csr_from_edges(
            [(1, d, x) for d, x in zip(c["dst_ids"], c["dst_w"])] +
            [(0, d, 1.0) for d in c["src_nbs"]], n_vertices=40)
        # vertex ids: s is relabelled 0?  keep the original ids instead:
        edges = [(c["src_id"], d, 1.0) for d in c["src_nbs"]]
        v = 35  # an id unused by karate (34 vertices)
        edges += [(v, d, x) for d, x in zip(c["dst_ids"], c["dst_w"])]
        rowptr, col, w = oracle.csr_from_edges(edges, n_vertices=40)
        pr = oracle.transition_probs(rowptr, col, w, c["src_id"], v, c["p"], c["q"])
        np.testing.assert_allclose(pr, c["implied"], rtol=0, atol=1e-12)
        n += 1
    assert n > 50
