import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "statistical: asserts a statistical tolerance (runs after "
                            "every deterministic test)")


# Order of the -m gpu run (the driver runs it with -x): the bit-exact parity tests against the
# reference's golden vectors and the oracle come first, then the API / edge-case / multi-rank
# tests, then the full-size property tests, and the statistical tests (chi-square, hogwild
# quality) last -- a statistical tolerance must never hide a deterministic parity test.
_ORDER = ["test_walk_gpu", "test_transformers_gpu", "test_wedge_gpu", "test_weighted_lanes_gpu", "test_margin_adversary_gpu", "test_alias_trim_fast_gpu",
          "test_sgns_gpu", "test_sgns_window_gpu", "test_edge_cases_gpu", "test_api_gpu",
          "test_indexer_gpu", "test_partitioned_gpu", "test_delta_sync_gpu", "test_multirank_gpu",
          "test_scale_props_gpu", "test_scale_cfg345_gpu", "test_sgns_model_size_gpu", "test_fast_unit_gpu",
          "test_sgns_batched_gpu", "test_sgns_parity_gpu"]
_STAT_MARK = "statistical"


def _rank(item):
    name = item.module.__name__.rsplit(".", 1)[-1] if item.module else ""
    base = _ORDER.index(name) if name in _ORDER else len(_ORDER) // 2
    # inside a file: tests marked statistical after the deterministic ones
    return (1 if item.get_closest_marker(_STAT_MARK) else 0, base)


def pytest_collection_modifyitems(config, items):
    """-m gpu tests are the parity tests proper and need a HIP device: skip them (rather than
    fail) on a GPU-less host; on the GPU box nothing is skipped.  Also fixes the order (above)."""
    import torch

    items.sort(key=_rank)  # stable: the order inside a file is kept
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import n2v_oracle

    n2v_oracle.build()
    return n2v_oracle
