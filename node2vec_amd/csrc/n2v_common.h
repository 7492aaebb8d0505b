// n2v_common.h -- shared device helpers for the gfx950 node2vec kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/n2v_hip.h"

#define N2V_WAVE 64

#define N2V_HIP_CHECK(expr)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return N2V_ELAUNCH;       \
  } while (0)

namespace n2v {

// Grid size of a persistent (grid-stride) kernel: CUs x resident blocks per CU of the
// current device (occupancy query is advisory; a larger grid would only queue blocks).
inline int64_t resident_blocks(const void *kernel, int block_threads, size_t dyn_lds) {
  int dev = 0, cus = 256, per_cu = 4;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block_threads, dyn_lds) == hipSuccess &&
      occ > 0)
    per_cu = occ;
  return (int64_t)cus * per_cu;
}

// Work distribution of a short single-step launch (n2v_partition_step): k items, a few per wave.
// One atomic per item on one counter costs ~12 ns each, serialised (118 k items: 1.4 ms, more than
// the work), so three quarters of the items are dealt statically, interleaved over the waves, and
// only the last quarter -- which evens out the differences in cost -- comes from the counter, two
// at a time.  Wave-uniform; `counter` must be zero at launch.
struct ItemQueue {
  int64_t k, n_waves, wave_id, per, j, base;
  uint32_t left;
  __device__ __forceinline__ ItemQueue(int64_t items, int waves_per_block)
      : k(items), n_waves((int64_t)gridDim.x * waves_per_block),
        wave_id((int64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6)), j(0), base(0), left(0) {
    per = (k - k / 4) / n_waves;
  }
  // next item of this wave, or -1 when there is none left
  __device__ __forceinline__ int64_t next(uint32_t *counter, int lane) {
    if (j < per) return wave_id + (j++) * n_waves;
    if (left == 0) {
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(counter, 2u);
      base = per * n_waves + (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      left = 2;
    }
    --left;
    const int64_t i = base++;
    return i < k ? i : -1;
  }
};

// splitmix64 finaliser; the uniform stream of DESIGN.md "RNG".
__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__host__ __device__ inline uint64_t walker_stream(uint64_t seed, uint64_t walk_key) {
  return mix64(seed ^ mix64(walk_key + 0x9E3779B97F4A7C15ULL));
}

// the two 32-bit uniforms of step `step` (randomwalk.py:336-337 replacement)
__host__ __device__ inline uint64_t step_bits(uint64_t h0, uint32_t step) {
  return mix64(h0 + ((uint64_t)step + 1ULL) * 0xD1B54A32D192ED03ULL);
}

// extra uniforms for rejection trials (fast mode only)
__host__ __device__ inline uint64_t trial_bits(uint64_t hstep, uint32_t trial) {
  return mix64(hstep ^ (((uint64_t)trial + 1ULL) * 0x8CB92BA72F3D8DD7ULL));
}

// pick = int(r1 * n), r1 = u1 / 2^32 (sampling_from_alias, randomwalk.py:95).  Below 2^21
// neighbours u1 * n < 2^53, the fp64 product is exact and the integer high word is the same
// number; for longer rows Python's product rounds once before int() truncates, so the same
// two fp64 operations are used.
__device__ inline int pick_index(uint32_t u1, int n) {
  if (n < (1 << 21)) return (int)__umulhi(u1, (uint32_t)n);
  return (int)(((double)u1 * (1.0 / 4294967296.0)) * (double)n);
}

// hop table entry (include/n2v_hip.h): one 16-byte load
__device__ __forceinline__ n2v_hop load_hop(const n2v_hop *p) {
  const int4 v = *reinterpret_cast<const int4 *>(p);
  n2v_hop h;
  h.col = v.x;
  h.classes = (uint32_t)v.y;
  h.row = (uint64_t)(uint32_t)v.z | ((uint64_t)(uint32_t)v.w << 32);
  return h;
}
__device__ __forceinline__ int64_t hop_row(const n2v_hop &h) { return (int64_t)(h.row & N2V_HOP_ROW_MASK); }
__device__ __forceinline__ int hop_deg(const n2v_hop &h) { return (int)(h.row >> N2V_HOP_DEG_SHIFT); }

__device__ inline int lane_id() { return __lane_id(); }

__device__ inline double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

__device__ inline double readfirstlane_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return __hiloint2double(hi, lo);
}

__device__ inline int64_t readfirstlane_i64(int64_t v) {
  int lo = (int)(uint32_t)v, hi = (int)(uint32_t)((uint64_t)v >> 32);
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ inline uint64_t readfirstlane_u64(uint64_t v) { return (uint64_t)readfirstlane_i64((int64_t)v); }

__device__ inline uint64_t ballot64(bool p) { return __ballot(p); }

// "x in a[0, m)" for a sorted row, by ONE lane (data-dependent trip count: lanes diverge)
__device__ __forceinline__ bool member_sorted_lane(const int32_t *a, int m, int32_t x) {
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < m && a[lo] == x;
}

// entry k of a wedge list.  The width (uint16 / uint32) is wave-uniform and every caller
// branches on it OUTSIDE its loops, each side doing plain typed loads: written as one
// `wide ? load32 : load16` expression the compiler turned the choice into a select of two
// loads, and the 32-bit one of a 16-bit table reads up to twice as far as the table is long
// (a memory fault once the table has its own allocation).
template <typename P>
__device__ __forceinline__ int wedge_at_t(const void *base, int64_t k) {
  return (int)reinterpret_cast<const P *>(base)[k];
}

// Entries of the ascending list a[0, cnt) below `pos` (its lower bound), by one lane.  A plain binary search
// is a chain of log2(cnt) DEPENDENT probes, and on the lists of a hub row (10^3 - 10^4 entries: graphs trimmed
// at the reference's own cap of 100 000) nearly every probe is a cache miss: the closed forms of a pairing
// there make 2 - 4 such searches, 10 - 25 k cycles in all (profiles/r7f_big_stats.log).  So a long range is
// cut by SEVEN evenly spaced probes at a time -- independent loads, one round trip -- to an eighth, and
// only the last 64 entries (two or three sectors) are searched by halving.
#ifndef N2V_LIST_KARY
#define N2V_LIST_KARY 1
#endif
#ifndef N2V_LIST_KARY_MIN
#define N2V_LIST_KARY_MIN 64  // ranges above this many entries are cut 8-ary; below, by halving
#endif
template <typename P>
__device__ __forceinline__ int list_lower_bound(const P *a, int cnt, int pos) {
  int lo = 0, hi = cnt;
#if N2V_LIST_KARY
  while (hi - lo > N2V_LIST_KARY_MIN) {
    const int step = (hi - lo) >> 3;
    int v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = (int)a[lo + (k + 1) * step];
    int below = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) below += v[k] < pos ? 1 : 0;
    // the pivots below `pos` are the first `below` of them (the list ascends)
    const int nlo = below == 0 ? lo : lo + below * step + 1;
    const int nhi = below == 7 ? hi : lo + (below + 1) * step;
    lo = nlo;
    hi = nhi;
  }
#endif
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)a[mid] < pos)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// A shared-position list as the routines of the pairing loop read it.  Plain: the ascending positions themselves.
// FOLDED (round 6; the lists of the edges into rows of T = n2v_graph.wedge_wide .. T + 65536 slots, whose positions do
// not fit 16 bits -- the reference's own trim cap is 100 000, constants.py:6): a position below T is stored as it is,
// one from T on minus T, in the same order, and `nlow` says how many entries lie below T -- so entry k is
// p[k] + (k >= nlow ? T : 0), both parts ascend, and a lower bound is ONE search in one of the two parts.  Such a list
// is 16-bit like every other, has a wedge slot like every other, and its steps run through the SAME instructions as
// everybody else's (a wave pays for every code path ONE of its lanes takes: the separate 32-bit instance of the step
// cost cfg 4 trimmed at 100 000 a sixth of its time at (0.5, 2) and three fifths at (3, 0.7), profiles/r11c_*).
// A raw pointer converts to the plain form, so callers that hold plain lists pass them as before.
template <typename P>
struct ListRef {
  const P *p;
  int nlow;  // entries [nlow, ..) carry + fold; 0x7fffffff: a plain list
  int fold;
  __host__ __device__ __forceinline__ ListRef(const P *q) : p(q), nlow(0x7fffffff), fold(0) {}
  __host__ __device__ __forceinline__ ListRef(const P *q, int nl, int f) : p(q), nlow(nl), fold(f) {}
  __device__ __forceinline__ int fix(int k, int raw) const { return k >= nlow ? raw + fold : raw; }
  __device__ __forceinline__ int operator[](int k) const { return fix(k, (int)p[k]); }
};

template <typename P>
__device__ __forceinline__ int list_lower_bound(const ListRef<P> &L, int cnt, int pos) {
  const P *a = L.p;
  int c = cnt, t = pos, add = 0;
  if (L.nlow < cnt) {  // a folded list with entries in both parts: `pos` says which part holds its lower bound
    if (pos < L.fold) {
      c = L.nlow;
    } else {
      a += L.nlow;
      c = cnt - L.nlow;
      t = pos - L.fold;
      add = L.nlow;
    }
  }
  return add + list_lower_bound<P>(a, c, t);
}

// is `pos` one of the (ascending) positions list[0, cnt)?  one lane
template <typename P>
__device__ __forceinline__ bool wedge_has_t(const void *base, int64_t off, int cnt, int pos) {
  const P *a = reinterpret_cast<const P *>(base) + off;
  const int lo = list_lower_bound<P>(a, cnt, pos);
  if (lo >= cnt) return false;
  return (int)a[lo] == pos;
}

// number of entries of list[0, cnt) below `pos` (its lower bound); found = pos is in the list
template <typename P>
__device__ __forceinline__ int wedge_lower_t(const void *base, int64_t off, int cnt, int pos,
                                             bool &found) {
  const P *a = reinterpret_cast<const P *>(base) + off;
  const int lo = list_lower_bound<P>(a, cnt, pos);
  found = false;
  if (lo < cnt) found = (int)a[lo] == pos;
  return lo;
}

// width of the wedge list a walker reads while it stands on a row of n entries (n2v_graph.wedge_wide:
// 0 uint16 everywhere, 1 uint32 everywhere, T >= 2 mixed -- uint32 for the rows of T entries or more)
__host__ __device__ __forceinline__ bool wedge_row_wide(int32_t wedge_wide, int64_t n) {
  return wedge_wide == 1 || (wedge_wide >= 2 && n >= (int64_t)wedge_wide);
}

__device__ __forceinline__ int wedge_lower(const void *base, int64_t off, int cnt, int pos, bool wide,
                                           bool &found) {
  if (wide) return wedge_lower_t<uint32_t>(base, off, cnt, pos, found);
  return wedge_lower_t<uint16_t>(base, off, cnt, pos, found);
}

__device__ __forceinline__ bool wedge_has(const void *base, int64_t off, int cnt, int pos, bool wide) {
  if (wide) return wedge_has_t<uint32_t>(base, off, cnt, pos);
  return wedge_has_t<uint16_t>(base, off, cnt, pos);
}

// ---- wedge slots (n2v_wedge_slots_build, include/n2v_hip.h): 16 halfwords per edge, loaded as two
// 16-byte words.  [0] return position, [1] list entries below it, then the list itself (<= 14
// entries) or, for a longer one, its 64-bit offset in wedge_pos at [4 .. 8) and eight pivots
// list[((k + 1) n) / 9] at [8 .. 16).
constexpr int kSlotShort = 14;
constexpr int kSlotPivots = 8;

__device__ __forceinline__ int slot_half(const int4 &a, const int4 &b, int k) {  // halfword k, k constant
  const int w = k >> 1;
  const uint32_t d = (uint32_t)(w == 0 ? a.x : w == 1 ? a.y : w == 2 ? a.z : w == 3 ? a.w
                              : w == 4 ? b.x : w == 5 ? b.y : w == 6 ? b.z : b.w);
  return (int)((k & 1) ? (d >> 16) : (d & 0xffffu));
}

// lower bound of `pick` in the list of a slot (entries below it; found = it is in the list).  A folded list (ListRef):
// entries from `nlow` on stand for their value + fold; nlow >= nM is a plain list.
__device__ __forceinline__ int slot_lower(const int4 &sa, const int4 &sb, int nM, int pick,
                                          const uint16_t *wedge_pos, bool &found, int nlow = 0x7fffffff,
                                          int fold = 0) {
  found = false;
  if (nM <= kSlotShort) {
    int lo = 0;
#pragma unroll
    for (int k = 0; k < kSlotShort; ++k) {
      const int raw = slot_half(sa, sb, k + 2);
      const int e = k >= nlow ? raw + fold : raw;
      const bool in = k < nM;
      lo += (in && e < pick) ? 1 : 0;
      found = found || (in && e == pick);
    }
    return lo;
  }
  const uint64_t off = (uint64_t)(uint32_t)sa.z | ((uint64_t)(uint32_t)sa.w << 32);
  const uint16_t *list = wedge_pos + off;
  int j = 0;  // pivots below pick
#pragma unroll
  for (int k = 0; k < kSlotPivots; ++k) {
    const int raw = slot_half(sa, sb, 8 + k);
    const int at = (int)(((int64_t)(k + 1) * nM) / 9);
    j += ((at >= nlow ? raw + fold : raw) < pick) ? 1 : 0;
  }
  // list[idx(j - 1)] < pick <= list[idx(j)], idx(k) = ((k + 1) nM) / 9, idx(-1) = -1, idx(8) = nM
  int lo = j == 0 ? 0 : (int)(((int64_t)j * nM) / 9) + 1;
  int hi = j == kSlotPivots ? nM : (int)(((int64_t)(j + 1) * nM) / 9);
  const int top = hi;
  int target = pick;
  if (nlow < nM) {  // folded: the part of the ninth that can hold the lower bound
    if (pick < fold) {
      hi = hi < nlow ? hi : nlow;
      lo = lo < hi ? lo : hi;
    } else {
      lo = lo > nlow ? lo : nlow;
      hi = hi > lo ? hi : lo;
      target = pick - fold;
    }
  }
  lo += list_lower_bound<uint16_t>(list + lo, hi - lo, target);
  if (lo < nM) {
    int e = 0;  // list[lo]: the pivot itself when the search ran to the end of its ninth
    if (lo == top && j < kSlotPivots) {
#pragma unroll
      for (int k = 0; k < kSlotPivots; ++k)
        if (k == j) e = slot_half(sa, sb, 8 + k);
    } else {
      e = (int)list[lo];
    }
    if (lo >= nlow) e += fold;
    found = e == pick;
  }
  return lo;
}

// entry i of the list of a slot (i < n), i not a compile-time constant
__device__ __forceinline__ int slot_entry(const int4 &sa, const int4 &sb, int n, int i,
                                          const uint16_t *wedge_pos) {
  if (n > kSlotShort)
    return (int)wedge_pos[((uint64_t)(uint32_t)sa.z | ((uint64_t)(uint32_t)sa.w << 32)) + (uint64_t)i];
  const int w = (i + 2) >> 1;  // dword 1 .. 7 of the slot
  const uint32_t d = (uint32_t)(w < 4 ? (w == 1 ? sa.y : (w == 2 ? sa.z : sa.w))
                                      : (w < 6 ? (w == 4 ? sb.x : sb.y) : (w == 6 ? sb.z : sb.w)));
  return (int)((i & 1) ? (d >> 16) : (d & 0xffffu));
}

__device__ inline int64_t wave_sum_i64(int64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
}  // namespace n2v
