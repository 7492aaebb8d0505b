"""The reference's row-level surface (randomwalk.py:17-153, :266-296, :343-349) as exported by
node2vec_amd.randomwalk: data carriers, wire format and the transformers that do no arithmetic.
(The numeric ones run on the GPU: tests/test_transformers_gpu.py.)"""
import pandas as pd
import pytest

from conftest import load_golden


def test_wire_format_matches_the_strings_the_reference_pins():
    from node2vec_amd.randomwalk import AliasProb, Neighbors, RandomPath

    g = load_golden("g8_wire.json")
    for c in g["neighbors"]:
        df = pd.DataFrame.from_dict({"dst": c["ids"], "weight": c["weights"]})
        for nb in (Neighbors((c["ids"], c["weights"])), Neighbors(c["code"]), Neighbors(df)):
            assert nb.dst_id == c["ids"] and nb.dst_wt == c["weights"]
            assert list(nb.items()) == list(zip(c["ids"], c["weights"]))
            assert nb.serialize() == c["code"]
            assert nb.as_pandas().equals(df)
    for c in g["alias_prob"]:
        df = pd.DataFrame.from_dict({"alias": c["alias"], "probs": c["probs"]})
        for ap in (AliasProb((c["alias"], c["probs"])), AliasProb(c["code"]), AliasProb(df)):
            assert ap.alias == c["alias"] and ap.probs == c["probs"]
            assert ap.serialize() == c["code"]
    for c in g["random_path"]:
        for rp in (RandomPath(c["path"]), RandomPath(c["code"])):
            assert rp.path == c["path"] and rp.last_edge == (c["path"][-2], c["path"][-1])
            assert rp.serialize() == c["code"] and str(rp) == str(c["path"])


def test_get_vertex_neighbors_initiate_and_to_path():
    from node2vec_amd.randomwalk import get_vertex_neighbors, initiate_random_walk, to_path

    g5 = load_golden("g5_next_step.json")
    df = pd.DataFrame.from_dict({"src": [3, 3, 3], "dst": [0, 1, 2], "weight": [1.0, 0.2, 1.4]})
    res = next(iter(get_vertex_neighbors(df)))  # tests/test_randomwalk.py:228-242
    assert sorted(res.keys()) == ["id", "neighbors"] and res["id"] == 3
    assert res["neighbors"] == load_golden("g8_wire.json")["neighbors"][0]["code"]
    rows = list(initiate_random_walk([{"id": 3, "neighbors": [0, 1, 2]}, {"id": 2, "neighbors": [3, 1]}], 3))
    assert rows == g5["initiate"]
    paths = [{"path": r["walk"]} for r in g5["to_path"]]
    assert list(to_path(paths)) == g5["to_path"]


def test_first_step_rule_of_path_extension():
    from node2vec_amd.randomwalk import RandomPath

    assert RandomPath([-2, 7])._extended(9).path == [7, 9]  # randomwalk.py:146-148
    assert RandomPath([3, 7])._extended(9).path == [3, 7, 9]
    assert RandomPath([-2, 7, 8])._extended(9).path == [-2, 7, 8, 9]


def test_numeric_transformers_fail_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from node2vec_amd.randomwalk import AliasProb, generate_alias_tables, next_step_random_walk

    with pytest.raises(RuntimeError):
        generate_alias_tables([0.5, 0.8, 1.0])
    with pytest.raises(RuntimeError):
        AliasProb(([1, 0], [0.6, 1.0])).sampling_from_alias(0.1, 0.2)
    with pytest.raises(RuntimeError):
        list(next_step_random_walk([{"src": -1, "path": [-1, 0], "src_neighbors": None,
                                     "dst_neighbors": ([1], [1.0])}], 1.0, 1.0, 1))
