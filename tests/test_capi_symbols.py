"""The C-ABI library loads on a GPU-less host and exports every symbol that
include/n2v_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "n2v_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(n2v_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for want in ("n2v_abi_version", "n2v_status_string", "n2v_device_count", "n2v_alias_build", "n2v_pivots_build",
                 "n2v_edge_classes_build", "n2v_walk", "n2v_trim_mark", "n2v_sgns_train",
                 "n2v_delta_ref_init", "n2v_delta_pack", "n2v_delta_apply", "n2v_edge_bias", "n2v_alias_draw",
                 "n2v_hops_build", "n2v_cum_index_build", "n2v_walk_uniforms", "n2v_wedge_build",
                 "n2v_mem_probe", "n2v_corpus_count", "n2v_corpus_index", "n2v_hops8_build",
                 "n2v_partition_step", "n2v_gather_rows", "n2v_gather_wedges",
                 "n2v_partition_route", "n2v_partition_group", "n2v_walk_ws",
                 "n2v_walk_workspace_bytes", "n2v_delta_reduce", "n2v_wedge_slots_build", "n2v_sgns_job_alpha", "n2v_rank_hops_build", "n2v_partition_forward",
                 "n2v_wedge_slots_fold", "n2v_edge_row_sums_build"):
        assert want in names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    from node2vec_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == _declared()
    lib.n2v_abi_version.restype = ctypes.c_int
    assert lib.n2v_abi_version() == _lib.ABI_VERSION == 15
    lib.n2v_status_string.restype = ctypes.c_char_p
    assert lib.n2v_status_string(-1) == b"invalid argument"


def test_ctypes_structs_match_header_layout():
    """sizeof / field order of the two structs passed by pointer"""
    from node2vec_amd import _lib

    assert ctypes.sizeof(_lib.Graph) == 18 * 8 + 6 * 8 + 4 * 4 + 4 * 8
    assert [f[0] for f in _lib.Graph._fields_] == ["n_vertices", "n_edges", "rowptr", "col", "w", "w64",
                                                    "slots", "pivots", "edge_classes", "hops", "wedge_off", "wedge_pos",
                                                    "wedge_wide", "reserved", "hops8", "hop8_col_bits", "hop8_row_bits",
                                                    "hop8_rowptr", "hop8_align_shift", "reserved2", "wedge_slots",
                                                    "rank_hops", "rank_of", "rank_vertex", "rank_head",
                                                    "rank_class_first", "rank_class_off", "rank_head_n",
                                                    "rank_classes", "rank_emit", "reserved3", "row_sums", "row_sums_p",
                                                    "row_sums_q", "row_sums_from", "reserved4"]
    # the header's field order, read from the header itself
    text = open(os.path.join(ROOT, "include", "n2v_hip.h")).read()
    body = text[text.index("typedef struct n2v_graph {"):text.index("} n2v_graph;")]
    fields = re.findall(r"\*?\s*\b(\w+);", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert fields == [f[0] for f in _lib.Graph._fields_]
    assert ctypes.sizeof(_lib.SgnsParams) == 3 * 8 + 6 * 4 + 8 + 4 * 4 + 8
    assert [f[0] for f in _lib.SgnsParams._fields_] == [
        "n_vocab", "sentence_base", "seed", "dim", "window", "negative", "alpha",
        "deterministic", "cum_index_bits", "cum_index", "max_waves", "batched", "window_cache",
        "hub_rows", "row_alpha"]
    body = text[text.index("typedef struct n2v_sgns_params {"):text.index("} n2v_sgns_params;")]
    fields = re.findall(r"\*?\s*\b(\w+);", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert fields == [f[0] for f in _lib.SgnsParams._fields_]


def test_partition_entry_points_validate_their_arguments():
    """argument errors of the partitioned-walking entry points come back as N2V_EINVAL (-1) before
    anything is launched (no GPU needed: nothing is launched); an empty batch is N2V_OK"""
    from node2vec_amd import _lib

    L = _lib.load()
    buf = (ctypes.c_int64 * 64)()
    p_ = ctypes.addressof(buf)

    def step(**kw):
        a = dict(rowptr=p_, col=p_, w=0, w64=0, lo=0, n_local=4, head=p_, head_cols=5, src_ptr=p_,
                 src_ids=p_, src_kind=0, k=1, p=0.5, q=2.0, seed=1, next_out=p_, edge_out=0, status=p_)
        a.update(kw)
        return L.n2v_partition_step(a["rowptr"], a["col"], a["w"], a["w64"], a["lo"], a["n_local"], a["head"],
                                    a["head_cols"], a["src_ptr"], a["src_ids"], a["src_kind"], a["k"], a["p"],
                                    a["q"], a["seed"], a["next_out"], a["edge_out"], a["status"], None)

    assert step(k=0) == 0  # nothing to do
    assert step(p=0.0) == -1 and step(q=0.0) == -1  # randomwalk.py:209-212
    assert step(k=-1) == -1 and step(n_local=-1) == -1 and step(head_cols=3) == -1
    assert step(src_kind=3) == -1 and step(src_kind=-1) == -1
    assert step(src_kind=2, w=p_) == -1  # N2V_SRC_WEDGES_AT: unit-weight parts only
    assert step(w=p_, w64=p_) == -1  # at most one weight array
    assert step(src_kind=1, w=p_) == -1  # wedge lists: unit-weight parts only
    assert step(head=0) == -1 and step(next_out=0) == -1 and step(status=0) == -1
    assert step(src_ptr=0) == -1  # q != 1: something must have travelled
    def forward(**kw):
        a = dict(head=p_, head_cols=5, next=p_, edge=p_, k=1, walk_length=10, bounds=p_, n_parts=2, carry=2,
                 ec=p_, off=p_, pos=p_, wide=0, box_head=p_, box_off=p_, box_words=p_, box_count=p_, cap=4,
                 wcap=4, log=p_, walks=0, valid=0, status=p_)
        a.update(kw)
        return L.n2v_partition_forward(a["head"], a["head_cols"], a["next"], a["edge"], a["k"], a["walk_length"],
                                       a["bounds"], a["n_parts"], a["carry"], a["ec"], a["off"], a["pos"],
                                       a["wide"], a["box_head"], a["box_off"], a["box_words"], a["box_count"],
                                       a["cap"], a["wcap"], a["log"], a["walks"], a["valid"], a["status"], None)

    def boxes(**kw):  # n2v_partition_forward_boxes: ragged mailboxes, the path records always to the log
        a = dict(k=1, starts=p_, log=p_, n_parts=2)
        a.update(kw)
        return L.n2v_partition_forward_boxes(p_, 5, p_, p_, a["k"], 10, p_, a["n_parts"], 2, p_, p_, p_, 0, p_, p_,
                                             p_, p_, a["starts"], a["log"], p_, None)

    assert boxes(k=0) == 0 and boxes(starts=0) == -1 and boxes(log=0) == -1 and boxes(n_parts=0) == -1
    assert forward(k=0) == 0
    assert forward(carry=1) == -1  # rows do not travel this way
    assert forward(carry=2, head_cols=4) == -1 and forward(k=-1) == -1 and forward(n_parts=0) == -1
    assert forward(log=0) == -1  # neither a log nor the output rows
    assert forward(carry=2, edge=0) == -1 and forward(carry=2, box_words=0) == -1 and forward(box_count=0) == -1
    assert L.n2v_partition_route(p_, 5, p_, 0, 0, 10, p_, 2, 0, p_, 0, 0, p_, p_, p_, p_, p_, None) == 0
    assert L.n2v_partition_route(p_, 3, p_, 0, 1, 10, p_, 2, 0, p_, 0, 0, p_, p_, p_, p_, p_, None) == -1
    assert L.n2v_partition_route(p_, 5, p_, 0, 1, 10, p_, 0, 0, p_, 0, 0, p_, p_, p_, p_, p_, None) == -1
    assert L.n2v_partition_route(p_, 5, p_, 0, 1, 10, p_, 2, 4, p_, 0, 0, p_, p_, p_, p_, p_, None) == -1
    assert L.n2v_partition_route(p_, 5, p_, 0, 1, 10, p_, 2, 2, p_, 0, p_, p_, p_, p_, p_, p_, None) == -1  # no edges
    assert L.n2v_partition_route(p_, 5, p_, 0, 1, 10, p_, 2, 1, 0, 0, 0, p_, p_, p_, p_, p_, None) == -1  # no rowptr
    assert L.n2v_gather_rows(p_, p_, p_, p_, 0, p_, None) == 0 and L.n2v_gather_rows(p_, p_, p_, p_, -1, p_, None) == -1
    assert L.n2v_gather_rows(0, p_, p_, p_, 1, p_, None) == -1
    assert L.n2v_gather_wedges(p_, p_, p_, 0, p_, p_, 0, p_, p_, 5, None) == 0
    assert L.n2v_gather_wedges(p_, p_, p_, 0, p_, p_, 1, p_, p_, 4, None) == -1
    assert L.n2v_gather_wedges(0, p_, p_, 0, p_, p_, 1, p_, p_, 5, None) == -1


def test_product_package_never_touches_the_oracle():
    """no file of node2vec_amd/ imports, loads or mentions oracle/ code paths"""
    pkg = os.path.join(ROOT, "node2vec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "n2v_oracle" not in text.replace("oracle/n2v_oracle", ""), f
                assert "import n2v_oracle" not in text, f


def test_round6_table_builders_validate_their_arguments():
    """n2v_wedge_slots_fold / n2v_edge_row_sums_build (ABI 15): argument errors come back as N2V_EINVAL (-1) before
    anything is launched (no GPU needed); nothing to do is N2V_OK"""
    from node2vec_amd import _lib

    L = _lib.load()
    buf = (ctypes.c_int64 * 64)()
    p_ = ctypes.addressof(buf) & ~31  # (32-byte aligned, inside the buffer's page: never dereferenced)

    def graph(**kw):
        g = _lib.Graph()
        g.n_vertices, g.n_edges = 4, 8
        g.rowptr = g.col = g.edge_classes = g.wedge_off = g.wedge_pos = p_
        g.wedge_wide = 65536
        for k, v in kw.items():
            setattr(g, k, v)
        return g

    def fold(g, off=p_, pos=p_, slots=p_):
        return L.n2v_wedge_slots_fold(g, off, pos, slots, None)

    assert fold(graph(n_edges=0)) == 0 and fold(graph(n_edges=-1)) == -1
    assert fold(graph(wedge_wide=0)) == -1 and fold(graph(wedge_wide=1)) == -1 and fold(graph(wedge_wide=70000)) == -1
    assert fold(graph(edge_classes=0)) == -1 and fold(graph(wedge_off=0)) == -1 and fold(graph(rowptr=0)) == -1
    assert fold(graph(), off=0) == -1 and fold(graph(), slots=0) == -1
    assert fold(graph(), pos=p_ + 64) == -1  # the folded copies go into the graph's own wedge_pos
    assert fold(graph(), slots=p_ + 8) == -1  # 32-byte slots

    def sums(g, p=3.0, q=0.7, edges=p_, k=1, out=p_):
        return L.n2v_edge_row_sums_build(g, p, q, edges, k, out, None)

    assert sums(graph(), k=0) == 0 and sums(graph(), k=-1) == -1
    assert sums(graph(), p=0.0) == -1 and sums(graph(), q=0.0) == -1  # randomwalk.py:214-217
    assert sums(graph(w=p_)) == -1 and sums(graph(w64=p_)) == -1  # unit weights only
    assert sums(graph(edge_classes=0)) == -1 and sums(graph(wedge_pos=0)) == -1
    assert sums(graph(), edges=0) == -1 and sums(graph(), out=0) == -1
    assert sums(graph(), p=1e-30) == -1  # 1/p outside 2^-20 .. 2^20 and not dyadic: no unit-weight kernel walks it
