"""ctypes binding of node2vec_amd/libn2v_hip.so (the C ABI of include/n2v_hip.h).

There is deliberately no fallback: a missing library or a missing GPU raises.
N2V_HIP_LIB (environment) names another build of the SAME library (a -DN2V_CHECK / -DN2V_STATS
diagnostic build, a timing variant of scripts/build_variants.sh): it must export every symbol of
include/n2v_hip.h and the same ABI version, or the import fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libn2v_hip.so")
ABI_VERSION = 15

OK, EINVAL, ELAUNCH, ENOGPU = 0, -1, -2, -3
ST_ZERODIV, ST_RANGE, ST_OVERFLOW = 1, 2, 4
WALK_EXACT, WALK_FAST = 0, 1
WIRE_F32, WIRE_BF16 = 0, 1

# every symbol include/n2v_hip.h declares
SYMBOLS = ("n2v_abi_version", "n2v_status_string", "n2v_device_count", "n2v_alias_build",
           "n2v_pivots_build", "n2v_edge_classes_build", "n2v_walk", "n2v_trim_mark",
           "n2v_sgns_train", "n2v_delta_ref_init", "n2v_delta_pack", "n2v_delta_apply",
           "n2v_edge_bias", "n2v_alias_draw", "n2v_hops_build", "n2v_cum_index_build", "n2v_walk_uniforms", "n2v_wedge_build",
           "n2v_mem_probe", "n2v_corpus_count", "n2v_corpus_index", "n2v_hops8_build",
           "n2v_partition_step", "n2v_gather_rows", "n2v_gather_wedges",
           "n2v_partition_route", "n2v_partition_group", "n2v_walk_ws", "n2v_walk_workspace_bytes",
           "n2v_delta_reduce", "n2v_wedge_slots_build", "n2v_sgns_job_alpha", "n2v_rank_hops_build", "n2v_partition_forward",
           "n2v_sgns_hogwild_waves", "n2v_walk_weighted_step", "n2v_partition_forward_boxes", "n2v_walk_weighted_keys",
           "n2v_wedge_slots_fold", "n2v_edge_row_sums_build")


class WeightedHubs(C.Structure):
    """struct n2v_weighted_hubs"""
    _fields_ = [("block0", C.c_void_p), ("sorted", C.c_void_p), ("prefix", C.c_void_p),
                ("min_slots", C.c_int32), ("lane_cut", C.c_int32)]


class Graph(C.Structure):
    """struct n2v_graph"""
    _fields_ = [("n_vertices", C.c_int64), ("n_edges", C.c_int64),
                ("rowptr", C.c_void_p), ("col", C.c_void_p), ("w", C.c_void_p),
                ("w64", C.c_void_p), ("slots", C.c_void_p), ("pivots", C.c_void_p),
                ("edge_classes", C.c_void_p), ("hops", C.c_void_p),
                ("wedge_off", C.c_void_p), ("wedge_pos", C.c_void_p), ("wedge_wide", C.c_int32),
                ("reserved", C.c_int32), ("hops8", C.c_void_p), ("hop8_col_bits", C.c_int32),
                ("hop8_row_bits", C.c_int32), ("hop8_rowptr", C.c_void_p),
                ("hop8_align_shift", C.c_int32), ("reserved2", C.c_int32),
                ("wedge_slots", C.c_void_p), ("rank_hops", C.c_void_p), ("rank_of", C.c_void_p),
                ("rank_vertex", C.c_void_p), ("rank_head", C.c_void_p), ("rank_class_first", C.c_void_p),
                ("rank_class_off", C.c_void_p), ("rank_head_n", C.c_int32), ("rank_classes", C.c_int32),
                ("rank_emit", C.c_int32), ("reserved3", C.c_int32),
                ("row_sums", C.c_void_p), ("row_sums_p", C.c_double), ("row_sums_q", C.c_double),
                ("row_sums_from", C.c_int32), ("reserved4", C.c_int32)]


class SgnsParams(C.Structure):
    """struct n2v_sgns_params"""
    _fields_ = [("n_vocab", C.c_int64), ("sentence_base", C.c_int64), ("seed", C.c_uint64),
                ("dim", C.c_int32), ("window", C.c_int32), ("negative", C.c_int32),
                ("alpha", C.c_float), ("deterministic", C.c_int32), ("cum_index_bits", C.c_int32),
                ("cum_index", C.c_void_p), ("max_waves", C.c_int32), ("batched", C.c_int32),
                ("window_cache", C.c_int32), ("hub_rows", C.c_int32), ("row_alpha", C.c_void_p)]


_lib = None


def load():
    """Load libn2v_hip.so; raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("N2V_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C node2vec_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "node2vec_amd has no CPU fallback.")
    L = C.CDLL(path)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{path} lacks {missing}: not a build of include/n2v_hip.h")
    L.n2v_abi_version.restype = C.c_int
    L.n2v_status_string.restype = C.c_char_p
    L.n2v_status_string.argtypes = [C.c_int]
    L.n2v_device_count.restype = C.c_int
    L.n2v_alias_build.restype = C.c_int
    if L.n2v_abi_version() != ABI_VERSION:
        raise ImportError(f"{path} has ABI version {L.n2v_abi_version()}, this package needs "
                          f"{ABI_VERSION}: rebuild it (make -C node2vec_amd/csrc)")
    L.n2v_alias_build.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_edge_classes_build.restype = C.c_int
    L.n2v_edge_classes_build.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_cum_index_build.restype = C.c_int
    L.n2v_cum_index_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.n2v_wedge_build.restype = C.c_int
    L.n2v_wedge_build.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_void_p]
    L.n2v_wedge_slots_build.restype = C.c_int
    L.n2v_wedge_slots_build.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p]
    L.n2v_wedge_slots_fold.restype = C.c_int
    L.n2v_wedge_slots_fold.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_edge_row_sums_build.restype = C.c_int
    L.n2v_edge_row_sums_build.argtypes = [C.POINTER(Graph), C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p]
    L.n2v_hops_build.restype = C.c_int
    L.n2v_hops_build.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_hops8_build.restype = C.c_int
    L.n2v_hops8_build.argtypes = [C.POINTER(Graph), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    L.n2v_partition_forward.restype = C.c_int
    L.n2v_partition_forward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_partition_forward_boxes.restype = C.c_int
    L.n2v_partition_forward_boxes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                              C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_rank_hops_build.restype = C.c_int
    L.n2v_rank_hops_build.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.n2v_pivots_build.restype = C.c_int
    L.n2v_pivots_build.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.n2v_walk.restype = C.c_int
    L.n2v_walk.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                           C.c_double, C.c_double, C.c_uint64, C.c_int32, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_walk_ws.restype = C.c_int
    L.n2v_walk_ws.argtypes = L.n2v_walk.argtypes[:-1] + [C.c_void_p, C.c_int64, C.c_void_p]
    L.n2v_walk_workspace_bytes.restype = C.c_int64
    L.n2v_walk_workspace_bytes.argtypes = [C.POINTER(Graph), C.c_int64, C.c_int32, C.c_int32,
                                           C.c_double, C.c_double, C.c_int32]
    L.n2v_trim_mark.restype = C.c_int
    L.n2v_trim_mark.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p,
                                C.c_void_p]
    L.n2v_walk_weighted_step.restype = C.c_int
    L.n2v_walk_weighted_step.argtypes = [C.POINTER(Graph), C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                         C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(WeightedHubs), C.c_void_p]
    L.n2v_walk_weighted_keys.restype = C.c_int
    L.n2v_walk_weighted_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p]
    L.n2v_sgns_hogwild_waves.restype = C.c_int64
    L.n2v_sgns_hogwild_waves.argtypes = [C.POINTER(SgnsParams), C.c_int64, C.c_int32]
    L.n2v_sgns_job_alpha.restype = C.c_int
    L.n2v_sgns_job_alpha.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_double,
                                     C.c_double, C.c_int64, C.c_void_p, C.c_void_p]
    L.n2v_sgns_train.restype = C.c_int
    L.n2v_sgns_train.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SgnsParams),
                                 C.c_void_p, C.c_void_p]
    L.n2v_delta_ref_init.restype = C.c_int
    L.n2v_delta_ref_init.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.n2v_delta_pack.restype = C.c_int
    L.n2v_delta_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_void_p]
    L.n2v_delta_reduce.restype = C.c_int
    L.n2v_delta_reduce.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.n2v_delta_apply.restype = C.c_int
    L.n2v_delta_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_int32, C.c_int64, C.c_void_p]
    L.n2v_edge_bias.restype = C.c_int
    L.n2v_edge_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double,
                                C.c_double, C.c_void_p, C.c_void_p]
    L.n2v_walk_uniforms.restype = C.c_int
    L.n2v_walk_uniforms.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    L.n2v_alias_draw.restype = C.c_int
    L.n2v_alias_draw.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    L.n2v_corpus_count.restype = C.c_int
    L.n2v_corpus_count.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p,
                                   C.c_void_p]
    L.n2v_partition_step.restype = C.c_int
    L.n2v_partition_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                     C.c_double, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.n2v_partition_route.restype = C.c_int
    L.n2v_partition_route.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.n2v_partition_group.restype = C.c_int
    L.n2v_partition_group.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.n2v_gather_wedges.restype = C.c_int
    L.n2v_gather_wedges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.n2v_gather_rows.restype = C.c_int
    L.n2v_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p]
    L.n2v_corpus_index.restype = C.c_int
    L.n2v_corpus_index.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                   C.c_void_p, C.c_void_p]
    L.n2v_mem_probe.restype = C.c_int
    L.n2v_mem_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(rc, what):
    """Map C status codes to the exceptions the reference raises."""
    if rc == OK:
        return
    msg = f"{what}: {load().n2v_status_string(rc).decode()}"
    if rc == EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg)


def check_status_word(word, what):
    if word & ST_ZERODIV:
        # generate_alias_tables: sum(weights) / n == 0 -> x / 0.0 (randomwalk.py:172-173)
        raise ZeroDivisionError(f"{what}: float division by zero (all neighbour weights are 0)")
    if word & ST_RANGE:
        raise ValueError(f"{what}: vertex id outside [0, n_vertices)")


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("node2vec_amd needs a HIP device (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def current_stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
