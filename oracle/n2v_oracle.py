"""ctypes binding of oracle/libn2v_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package node2vec_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, EINVAL, EZERODIV, ENOMEM = 0, -1, -2, -3


class CSR(C.Structure):
    _fields_ = [("n_vertices", C.c_int64), ("rowptr", C.c_void_p),
                ("col", C.c_void_p), ("w", C.c_void_p), ("w64", C.c_void_p)]


def _csr(rowptr, col, w):
    """(struct, keep-alive tuple): w None = unit weights, float64 = fp64 storage, else fp32"""
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    col = np.ascontiguousarray(col, np.int32)
    w32 = w64 = None
    if w is not None:
        w = np.asarray(w)
        if w.dtype == np.float64:
            w64 = np.ascontiguousarray(w)
        else:
            w32 = np.ascontiguousarray(w, np.float32)
    g = CSR(len(rowptr) - 1, _p(rowptr).value, _p(col).value,
            None if w32 is None else _p(w32).value, None if w64 is None else _p(w64).value)
    return g, (rowptr, col, w32, w64)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libn2v_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.n2v_oracle_alias_tables.restype = C.c_int
        L.n2v_oracle_edge_alias_tables.restype = C.c_int
        L.n2v_oracle_sampling_from_alias.restype = C.c_int64
        L.n2v_oracle_sampling_from_alias_wiki.restype = C.c_int64
        L.n2v_oracle_path_append.restype = C.c_int
        L.n2v_oracle_random_walk.restype = C.c_int
        L.n2v_oracle_transition_probs.restype = C.c_int
        L.n2v_oracle_edge_classes.restype = C.c_int
        L.n2v_oracle_sgns_train.restype = C.c_int64
        L.n2v_oracle_sgns_train_batched.restype = C.c_int64
        _LIB = L
    return _LIB


def _raise(rc):
    if rc == EINVAL:
        raise ValueError("oracle: invalid argument")
    if rc == EZERODIV:
        raise ZeroDivisionError("oracle: division by zero")
    if rc != OK:
        raise RuntimeError(f"oracle: status {rc}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def alias_tables(weights):
    w = np.ascontiguousarray(weights, dtype=np.float64)
    n = len(w)
    alias = np.zeros(max(n, 1), np.int32)
    probs = np.zeros(max(n, 1), np.float64)
    _raise(lib().n2v_oracle_alias_tables(_p(w), C.c_int64(n), _p(alias), _p(probs)))
    return alias[:n].tolist(), probs[:n].tolist()


def edge_alias_tables(src_id, src_nbs, dst_ids, dst_w, p=1.0, q=1.0):
    nb = np.ascontiguousarray(sorted(src_nbs), dtype=np.int32)
    ids = np.ascontiguousarray(dst_ids, dtype=np.int32)
    w = np.ascontiguousarray(dst_w, dtype=np.float64)
    n = max(len(ids), 1)
    alias = np.zeros(n, np.int32)
    probs = np.zeros(n, np.float64)
    _raise(lib().n2v_oracle_edge_alias_tables(
        C.c_int64(src_id), _p(nb), C.c_int64(len(nb)), _p(ids), _p(w),
        C.c_int64(len(ids)), C.c_int64(len(w)), C.c_double(p), C.c_double(q),
        _p(alias), _p(probs)))
    return alias[:len(ids)].tolist(), probs[:len(ids)].tolist()


def sampling_from_alias(alias, probs, r1, r2):
    a = np.ascontiguousarray(alias, dtype=np.int32)
    pr = np.ascontiguousarray(probs, dtype=np.float64)
    return int(lib().n2v_oracle_sampling_from_alias(_p(a), _p(pr), C.c_int64(len(a)),
                                                    C.c_double(r1), C.c_double(r2)))


def sampling_from_alias_wiki(alias, probs, r1):
    a = np.ascontiguousarray(alias, dtype=np.int32)
    pr = np.ascontiguousarray(probs, dtype=np.float64)
    return int(lib().n2v_oracle_sampling_from_alias_wiki(_p(a), _p(pr), C.c_int64(len(a)),
                                                         C.c_double(r1)))


def path_append(path, dst_nbs, alias, probs, r1, r2=None):
    buf = np.zeros(len(path) + 1, np.int64)
    buf[:len(path)] = path
    ln = C.c_int64(len(path))
    nb = np.ascontiguousarray(dst_nbs, dtype=np.int32)
    a = np.ascontiguousarray(alias, dtype=np.int32)
    pr = np.ascontiguousarray(probs, dtype=np.float64)
    _raise(lib().n2v_oracle_path_append(_p(buf), C.byref(ln), _p(nb), _p(a), _p(pr),
                                        C.c_int64(len(a)), C.c_double(r1),
                                        C.c_int(r2 is not None),
                                        C.c_double(0.0 if r2 is None else r2)))
    return buf[:ln.value].tolist()


def uniform_bits(seed, key, step):
    u1, u2 = C.c_uint32(), C.c_uint32()
    lib().n2v_oracle_uniform_bits(C.c_uint64(seed), C.c_uint64(key), C.c_uint32(step),
                                  C.byref(u1), C.byref(u2))
    return u1.value, u2.value


def csr_from_edges(edges, n_vertices=None):
    """edges: iterable of (src, dst, weight).  Stable sort by (src, dst), as the
    reference's partition(by=src, presort=dst) (fugue.py:130)."""
    e = list(edges)
    src = np.array([x[0] for x in e], dtype=np.int64)
    dst = np.array([x[1] for x in e], dtype=np.int64)
    # the reference carries Python floats (fp64): keep them, and use the 4-byte storage form
    # only when it holds the very same values (as DeviceGraph.from_edges does on the device)
    w = np.array([x[2] for x in e], dtype=np.float64)
    if np.array_equal(w.astype(np.float32).astype(np.float64), w):
        w = w.astype(np.float32)
    if n_vertices is None:
        n_vertices = int(max(src.max(initial=-1), dst.max(initial=-1)) + 1)
    order = np.lexsort((dst, src))  # stable: last key is primary
    src, dst, w = src[order], dst[order], w[order]
    rowptr = np.zeros(n_vertices + 1, np.int64)
    np.add.at(rowptr, src + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int64)
    return rowptr, dst.astype(np.int32), w


def random_walk(rowptr, col, w, start_ids, num_walks, walk_length, p, q, seed, n_threads=1):
    g, _keep = _csr(rowptr, col, w)
    start = np.ascontiguousarray(start_ids, np.int32)
    total = len(start) * num_walks
    walks = np.full((total, walk_length + 1), -1, np.int32)
    valid = np.zeros(total, np.uint8)
    _raise(lib().n2v_oracle_random_walk(
        C.byref(g), _p(start), C.c_int64(len(start)), C.c_int32(num_walks),
        C.c_int32(walk_length), C.c_double(p), C.c_double(q), C.c_uint64(seed),
        _p(walks), _p(valid), C.c_int32(n_threads)))
    return walks, valid.astype(bool)


def transition_probs(rowptr, col, w, s, v, p, q):
    g, _keep = _csr(rowptr, col, w)
    n = int(rowptr[v + 1] - rowptr[v])
    out = np.zeros(max(n, 1), np.float64)
    _raise(lib().n2v_oracle_transition_probs(C.byref(g), C.c_int64(s), C.c_int64(v),
                                             C.c_double(p), C.c_double(q), _p(out)))
    return out[:n]


def edge_classes(rowptr, col):
    g, _keep = _csr(rowptr, col, None)
    out = np.zeros(max(len(col), 1), np.uint32)
    _raise(lib().n2v_oracle_edge_classes(C.byref(g), _p(out)))
    return out[:len(col)]


def sgns_train(walks_idx, syn0, syn1neg, cum_table, sample_int, exp_table, n_vocab,
               sentence_base, seed, dim, window, negative, alpha, batched=False):
    """oracle/n2v_oracle_sgns.c: trains in place (syn0, syn1neg float32 C-contiguous),
    rows in order on one thread.  Returns the number of pairs trained."""
    walks_idx = np.ascontiguousarray(walks_idx, np.int32)
    assert syn0.dtype == np.float32 and syn0.flags.c_contiguous
    assert syn1neg.dtype == np.float32 and syn1neg.flags.c_contiguous
    cum = np.ascontiguousarray(cum_table).view(np.uint32)
    si = None if sample_int is None else np.ascontiguousarray(sample_int).view(np.uint32)
    et = np.ascontiguousarray(exp_table, np.float32)
    fn = lib().n2v_oracle_sgns_train_batched if batched else lib().n2v_oracle_sgns_train
    n = fn(
        _p(walks_idx), C.c_int64(walks_idx.shape[0]), C.c_int32(walks_idx.shape[1]),
        _p(syn0), _p(syn1neg), _p(cum), None if si is None else _p(si), _p(et),
        C.c_int64(n_vocab), C.c_int64(sentence_base), C.c_uint64(seed), C.c_int32(dim),
        C.c_int32(window), C.c_int32(negative), C.c_float(alpha))
    if n < 0:
        raise ValueError("oracle sgns: invalid argument")
    return int(n)


def trim_mark(rowptr, max_out_degree, seed):
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    keep = np.zeros(int(rowptr[-1]), np.uint8)
    _raise(lib().n2v_oracle_trim_mark(_p(rowptr), C.c_int64(len(rowptr) - 1),
                                      C.c_int64(max_out_degree), C.c_uint64(seed), _p(keep)))
    return keep.astype(bool)
