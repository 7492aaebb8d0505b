// n2v_walk_wedge2.hip -- K2 exact mode, biased, dyadic p and q, unit-weight graph with all three
// per-edge tables: the kernel of n2v_walk_wedge.hip with the replay code taken OUT of the hot loop.
//
// Same contract, same bits (reference randomwalk.py:86-99, :157-232; fugue.py:137-153).  In
// n2v_walk_wedge.hip a lane whose closed form declines (a tie or a thin margin: fp64 rounding
// decides the draw, ~1 % of the steps at p = 0.5, q = 2) replays the pairing loop on the spot: the
// other 63 walkers of its wave wait for it at about every second wave-step, and the replay routines
// cost every step their registers (80 VGPRs + 92..188 bytes of scratch per lane at 6 waves/SIMD).
// Here the work is cut in passes over a list of 32-byte walker records (the workspace of
// n2v_walk_ws, include/n2v_hip.h):
//   init     one record per walker {row, previous vertex, vertex, class counts of the edge walked,
//            row pointer | degree, edge | steps done}; path position 0 is written
//   main     persistent waves; a lane takes a record, walks it with the quick exits and the closed
//            forms ONLY, and when a closed form declines stores the record back ("parked") and takes
//            the next one from a small per-wave ring in LDS -- no lane waits for another lane's
//            replay, no replay code in the kernel
//   resolve  one lane per parked record: that ONE step with the replays (pair_listed), path word
//            written, record appended to the next list
// main / resolve alternate a fixed number of rounds (parked walkers: 59 %, 22 %, 6 %, 1.4 % of the
// batch at cfg 4); whatever is left is finished by `finish`, the one-launch shape with inline
// replays, started from the records.  Nothing synchronises with the host: list sizes stay on the
// device.  Walkers are independent and the uniforms are counter-based (DESIGN.md "RNG"), so the
// order in which steps are taken does not matter.
#include "n2v_wedge_step.h"

namespace n2v {

constexpr int kW2Threads = 256;
constexpr int kW2Ring = 32;  // records staged per wave
#ifndef N2V_W2_WAVES
#define N2V_W2_WAVES 7
#endif
constexpr uint32_t kRecDone = 0xffffffu;  // "steps done" field of a record that needs nothing more
constexpr int kW2Counters = 256;          // uint32 words at the head of the workspace

// a walker between passes (32 bytes: two 16-byte accesses)
struct WalkRec {
  uint32_t r;    // row of walks_out = start index * num_walks + ordinal - 1
  int32_t s;     // previous vertex, -1 before the first step
  int32_t v;     // current vertex
  uint32_t ec;   // class counts of edge (s -> v), edge_classes layout
  uint64_t row;  // rowptr[v] | degree(v) << 40 (the hop table's form)
  uint64_t es;   // index of edge (s -> v) | steps done << 40 (kRecDone: finished)
};
static_assert(sizeof(WalkRec) == 32, "two 16-byte accesses");

__device__ __forceinline__ WalkRec load_rec(const WalkRec *p) {
  const int4 a = reinterpret_cast<const int4 *>(p)[0];
  const int4 b = reinterpret_cast<const int4 *>(p)[1];
  WalkRec w;
  w.r = (uint32_t)a.x;
  w.s = a.y;
  w.v = a.z;
  w.ec = (uint32_t)a.w;
  w.row = (uint64_t)(uint32_t)b.x | ((uint64_t)(uint32_t)b.y << 32);
  w.es = (uint64_t)(uint32_t)b.z | ((uint64_t)(uint32_t)b.w << 32);
  return w;
}
__device__ __forceinline__ void store_rec(WalkRec *p, const WalkRec &w) {
  reinterpret_cast<int4 *>(p)[0] = make_int4((int)w.r, w.s, w.v, (int)w.ec);
  reinterpret_cast<int4 *>(p)[1] = make_int4((int)(uint32_t)w.row, (int)(uint32_t)(w.row >> 32),
                                             (int)(uint32_t)w.es, (int)(uint32_t)(w.es >> 32));
}
__device__ __forceinline__ void store_done(WalkRec *p) {
  p->es = (uint64_t)kRecDone << 40;
}

// ---- init: one record per walker (initiate_random_walk, randomwalk.py:279-296) ----------------
__global__ __launch_bounds__(256) void wedge2_init_kernel(
    const int64_t *__restrict__ rowptr, int64_t n_vertices, const int32_t *__restrict__ start_ids,
    int64_t total, int32_t num_walks, int32_t walk_length, WalkRec *__restrict__ list,
    uint32_t *__restrict__ n_list, int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out,
    uint32_t *__restrict__ status) {
  const int L1 = walk_length + 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_list = (uint32_t)total;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < total;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t start = start_ids[r / num_walks];
    bool alive = true;
    int64_t vb = 0;
    int n = 0;
    if (start < 0 || (int64_t)start >= n_vertices) {
      atomicOr(status, N2V_ST_RANGE);
      alive = false;
    } else {
      vb = rowptr[start];
      n = (int)(rowptr[start + 1] - vb);
      alive = n > 0;  // fugue.py:132
    }
    int32_t *row = walks_out + r * (int64_t)L1;
    WalkRec w;
    w.r = (uint32_t)r;
    w.s = -1;
    w.v = start;
    w.ec = 0;
    w.row = (uint64_t)vb | ((uint64_t)(uint32_t)n << N2V_HOP_DEG_SHIFT);
    w.es = 0;
    if (!alive) {  // no such vertex / no out-edges: the row is all -1, like the other kernels
      for (int t = 0; t < L1; ++t) row[t] = -1;
      valid_out[r] = 0;
      w.es = (uint64_t)kRecDone << 40;
    } else {
      row[0] = start;
      if (walk_length == 0) {
        valid_out[r] = 1;
        w.es = (uint64_t)kRecDone << 40;
      }
    }
    store_rec(list + r, w);
  }
}

// ---- main: quick exits and closed forms only; a declined step parks the walker ---------------
template <int kMode>
__global__ __launch_bounds__(kW2Threads, N2V_W2_WAVES) void wedge2_main_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int32_t num_walks, int32_t walk_length,
    double q, UnitConsts K, uint64_t seed, WalkRec *__restrict__ list,
    const uint32_t *__restrict__ n_list, uint32_t *__restrict__ dyn_counter,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  __shared__ int32_t path_tile[16][kW2Threads];  // word k of thread t at [k][t]
  __shared__ int4 ring_all[kW2Threads / 64][kW2Ring][3];  // record (2 x 16 B) + walker stream
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  int4(*ring)[3] = ring_all[tid >> 6];
  const int L1 = walk_length + 1;
  const StepFlags F = step_flags(g, K, q);
  const bool base_aligned = (reinterpret_cast<uintptr_t>(walks_out) & 63u) == 0;

  // the list is dealt in contiguous slices, one per wave (no atomics); the last eighth is handed
  // out in ring-sized chunks from a counter so that the waves finish together
  const int64_t n_items = (int64_t)*n_list;
  const int64_t n_waves = (int64_t)gridDim.x * (kW2Threads / 64);
  const int64_t wave_id = (int64_t)blockIdx.x * (kW2Threads / 64) + (tid >> 6);
  const int64_t per = (n_items - n_items / 8) / n_waves;
  const int64_t n_static = per * n_waves;
  int64_t cur = wave_id * per;
  const int64_t slice_end = cur + per;
  bool dyn_done = n_static >= n_items;
  int ring_head = 0, ring_count = 0;
  int64_t ring_item0 = 0;

  // per-lane walker
  bool walking = false;
  int64_t item = 0, w0 = 0, vb = 0, e_prev = 0;
  int32_t s = -1, v = -1;
  uint32_t ec_prev = 0, r = 0;
  uint64_t h0 = 0;
  int n = 0, step = 0, lo = 0;

  auto flush = [&](int64_t a) {  // words [sector(a) + lo, a] are complete: store them
    const int k = (int)(a & 15);
    int32_t *sec = walks_out + (a & ~(int64_t)15);
    if (lo == 0 && k == 15 && base_aligned) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        reinterpret_cast<int4 *>(sec)[u] =
            make_int4(path_tile[4 * u][tid], path_tile[4 * u + 1][tid], path_tile[4 * u + 2][tid],
                      path_tile[4 * u + 3][tid]);
    } else {
      for (int kk = lo; kk <= k; ++kk) sec[kk] = path_tile[kk][tid];
    }
    lo = 0;
  };
  auto emit = [&](int pos, int32_t x) {  // path position pos of the current walker
    const int64_t a = w0 + pos;
    path_tile[(int)(a & 15)][tid] = x;
    if ((a & 15) == 15 || pos == walk_length) flush(a);
  };

  for (;;) {
    // ---- lanes without a walker take the next records of the wave's ring
    const uint64_t free_mask = ballot64(!walking);
    if (free_mask != 0ull) {
      if (ring_count == 0) {  // refill the ring (wave-uniform)
        int64_t base = 0;
        int cnt = 0;
        if (cur < slice_end) {
          base = cur;
          cnt = (int)((slice_end - cur) < kW2Ring ? (slice_end - cur) : kW2Ring);
          cur += cnt;
        } else if (!dyn_done) {
          uint32_t t = 0;
          if (lane == 0) t = atomicAdd(dyn_counter, (uint32_t)kW2Ring);
          base = n_static + (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
          if (base >= n_items) {
            dyn_done = true;
          } else {
            cnt = (int)((n_items - base) < kW2Ring ? (n_items - base) : kW2Ring);
          }
        }
        if (cnt > 0) {
          if (lane < cnt) {
            const int4 a = reinterpret_cast<const int4 *>(list + base + lane)[0];
            const int4 b = reinterpret_cast<const int4 *>(list + base + lane)[1];
            // the walker's uniform stream (keyed by start vertex and ordinal), once per staging
            const uint32_t rr = (uint32_t)a.x;
            const int32_t start = start_ids[rr / (uint32_t)num_walks];
            const uint64_t hh = walker_stream(
                seed, (uint64_t)(uint32_t)start * (uint64_t)num_walks + (uint64_t)(rr % (uint32_t)num_walks));
            ring[lane][0] = a;
            ring[lane][1] = b;
            ring[lane][2] = make_int4((int)(uint32_t)hh, (int)(uint32_t)(hh >> 32), 0, 0);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          ring_item0 = base;
          ring_head = 0;
          ring_count = cnt;
        }
      }
      if (ring_count > 0) {
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u));
        if (!walking && rank < ring_count) {
          const int e = ring_head + rank;
          const int4 a = ring[e][0], b = ring[e][1], c = ring[e][2];
          item = ring_item0 + e;
          const uint64_t row = (uint64_t)(uint32_t)b.x | ((uint64_t)(uint32_t)b.y << 32);
          const uint64_t es = (uint64_t)(uint32_t)b.z | ((uint64_t)(uint32_t)b.w << 32);
          const uint32_t st = (uint32_t)(es >> 40);
          if (st != kRecDone) {
            r = (uint32_t)a.x;
            s = a.y;
            v = a.z;
            ec_prev = (uint32_t)a.w;
            vb = (int64_t)(row & N2V_HOP_ROW_MASK);
            n = (int)(row >> N2V_HOP_DEG_SHIFT);
            e_prev = (int64_t)(es & N2V_HOP_ROW_MASK);
            step = (int)st;
            h0 = (uint64_t)(uint32_t)c.x | ((uint64_t)(uint32_t)c.y << 32);
            w0 = (int64_t)r * (int64_t)L1;
            lo = (int)((w0 + step + 1) & 15);  // positions 0 .. step are in memory already
            walking = true;
          }
        }
        const int want = __popcll(free_mask);
        const int taken = want < ring_count ? want : ring_count;
        ring_head += taken;
        ring_count -= taken;
      }
    }
    if (ballot64(walking) == 0ull) {
      if (ring_count == 0 && cur >= slice_end && dyn_done) break;
      continue;
    }
    if (!walking) continue;

    // ---- one step
    const uint64_t bits = step_bits(h0, (uint32_t)step);
    const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
    n2v_hop h;
    int idx;
    if (s >= 0) {
      idx = wedge_step<kMode, true, false>(g, K, F, u1, u2, s, vb, n, e_prev, ec_prev, h, nullptr, lane, status);
    } else {  // first step: generate_alias_tables of unit weights is the uniform draw (:320-321)
      idx = pick_index(u1, n);
      h = load_hop(g.hops + vb + idx);
    }
    if (idx < 0) {
      // parked: the record goes back with the steps done so far, the words of the open sector
      // that belong to this row are stored, and the lane is free
      WalkRec w;
      w.r = r;
      w.s = s;
      w.v = v;
      w.ec = ec_prev;
      w.row = (uint64_t)vb | ((uint64_t)(uint32_t)n << N2V_HOP_DEG_SHIFT);
      w.es = (uint64_t)e_prev | ((uint64_t)(uint32_t)step << 40);
      store_rec(list + item, w);
      const int64_t a = w0 + step;
      if ((a & 15) != 15) flush(a);  // (lo > a & 15 right after a resume: nothing pending)
      walking = false;
      continue;
    }
    const int32_t x = h.col;
    emit(step + 1, x);
    e_prev = vb + idx;
    ec_prev = h.classes;
    s = v;
    v = x;
    ++step;
    if (step < walk_length) {
      vb = hop_row(h);
      n = hop_deg(h);
      if (n == 0) {  // fugue.py:147: the walker vanishes at a sink, the rest of its row is -1
        for (int tt = step + 1; tt < L1; ++tt) emit(tt, -1);
        valid_out[r] = 0;
        store_done(list + item);
        walking = false;
      }
    } else {
      valid_out[r] = 1;
      store_done(list + item);
      walking = false;
    }
  }
}

// ---- resolve: the parked step of every record of `list`, replays included; records that go on
// are appended to `next` (one atomic per 1024 records)
template <int kMode>
__global__ __launch_bounds__(kW2Threads) void wedge2_resolve_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int32_t num_walks, int32_t walk_length,
    double q, UnitConsts K, uint64_t seed, const WalkRec *__restrict__ list,
    const uint32_t *__restrict__ n_list, WalkRec *__restrict__ next, uint32_t *__restrict__ n_next,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  constexpr int kPer = 4;  // records per thread and pass
  __shared__ uint32_t stage_all[kW2Threads / 64][16 * 32];  // 2 KB per wave (lane_case_a)
  __shared__ uint32_t wave_cnt[kW2Threads / 64];
  __shared__ uint32_t block_base;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  uint32_t *stage = stage_all[tid >> 6];
  const int L1 = walk_length + 1;
  const StepFlags F = step_flags(g, K, q);
  const int64_t n_items = (int64_t)*n_list;
  const int64_t chunk = (int64_t)kW2Threads * kPer;
  for (int64_t c0 = (int64_t)blockIdx.x * chunk; c0 < n_items; c0 += (int64_t)gridDim.x * chunk) {
    WalkRec out[kPer];
    bool go[kPer];
#pragma unroll 1
    for (int u = 0; u < kPer; ++u) {
      const int64_t i = c0 + (int64_t)u * kW2Threads + tid;
      go[u] = false;
      if (i >= n_items) continue;
      const WalkRec w = load_rec(list + i);
      const uint32_t st = (uint32_t)(w.es >> 40);
      if (st == kRecDone) continue;
      const int64_t vb = (int64_t)(w.row & N2V_HOP_ROW_MASK);
      const int n = (int)(w.row >> N2V_HOP_DEG_SHIFT);
      const int64_t e_prev = (int64_t)(w.es & N2V_HOP_ROW_MASK);
      const int32_t start = start_ids[w.r / (uint32_t)num_walks];
      const uint64_t h0 = walker_stream(
          seed, (uint64_t)(uint32_t)start * (uint64_t)num_walks + (uint64_t)(w.r % (uint32_t)num_walks));
      const uint64_t bits = step_bits(h0, st);
      const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
      n2v_hop h;
      int idx;
      if (w.s >= 0) {
        idx = wedge_step<kMode, false, false>(g, K, F, u1, u2, w.s, vb, n, e_prev, w.ec, h, stage, lane, status);
      } else {
        idx = pick_index(u1, n);
        h = load_hop(g.hops + vb + idx);
      }
      const int32_t x = h.col;
      int32_t *row = walks_out + (int64_t)w.r * (int64_t)L1;
      const int done = (int)st + 1;
      row[done] = x;
      if (done == walk_length) {
        valid_out[w.r] = 1;
      } else if (hop_deg(h) == 0) {  // fugue.py:147
        for (int tt = done + 1; tt < L1; ++tt) row[tt] = -1;
        valid_out[w.r] = 0;
      } else {
        out[u].r = w.r;
        out[u].s = w.v;
        out[u].v = x;
        out[u].ec = h.classes;
        out[u].row = h.row;
        out[u].es = (uint64_t)(vb + idx) | ((uint64_t)(uint32_t)done << 40);
        go[u] = true;
      }
    }
    // append: count per thread, per wave, per block; one atomic per block and pass
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) mine += go[u] ? 1u : 0u;
    uint32_t incl = mine;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wave_cnt[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
      uint32_t tot = 0;
      for (int wv = 0; wv < kW2Threads / 64; ++wv) {
        const uint32_t c = wave_cnt[wv];
        wave_cnt[wv] = tot;
        tot += c;
      }
      block_base = tot ? atomicAdd(n_next, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = block_base + wave_cnt[tid >> 6] + (incl - mine);
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (go[u]) store_rec(next + pos++, out[u]);
    __syncthreads();  // wave_cnt / block_base are reused by the next pass
  }
}

// ---- finish: every record of `list` to the end of its walk, replays inline (the shape of
// walk_exact_wedge_kernel, started from records).  Runs on whatever the rounds left over.
template <int kMode>
__global__ __launch_bounds__(kW2Threads, 4) void wedge2_finish_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int32_t num_walks, int32_t walk_length,
    double q, UnitConsts K, uint64_t seed, const WalkRec *__restrict__ list,
    const uint32_t *__restrict__ n_list, uint32_t *__restrict__ counter,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  __shared__ uint32_t stage_all[kW2Threads / 64][16 * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  uint32_t *stage = stage_all[tid >> 6];
  const int L1 = walk_length + 1;
  const StepFlags F = step_flags(g, K, q);
  const int64_t n_items = (int64_t)*n_list;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(counter, 64u);
    const int64_t base = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (base >= n_items) break;
    const int64_t i = base + lane;
    bool walking = false;
    WalkRec w;
    w.r = 0; w.s = -1; w.v = -1; w.ec = 0; w.row = 0; w.es = 0;
    if (i < n_items) {
      w = load_rec(list + i);
      walking = (uint32_t)(w.es >> 40) != kRecDone;
    }
    int64_t vb = (int64_t)(w.row & N2V_HOP_ROW_MASK), e_prev = (int64_t)(w.es & N2V_HOP_ROW_MASK);
    int n = (int)(w.row >> N2V_HOP_DEG_SHIFT), step = (int)(uint32_t)(w.es >> 40);
    int32_t s = w.s, v = w.v;
    uint32_t ec_prev = w.ec;
    uint64_t h0 = 0;
    if (walking) {
      const int32_t start = start_ids[w.r / (uint32_t)num_walks];
      h0 = walker_stream(seed, (uint64_t)(uint32_t)start * (uint64_t)num_walks +
                                   (uint64_t)(w.r % (uint32_t)num_walks));
    }
    int32_t *row = walks_out + (int64_t)w.r * (int64_t)L1;
    while (ballot64(walking) != 0ull) {
      if (!walking) continue;
      const uint64_t bits = step_bits(h0, (uint32_t)step);
      const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
      n2v_hop h;
      int idx;
      if (s >= 0) {
        idx = wedge_step<kMode, false, false>(g, K, F, u1, u2, s, vb, n, e_prev, ec_prev, h, stage, lane, status);
      } else {
        idx = pick_index(u1, n);
        h = load_hop(g.hops + vb + idx);
      }
      const int32_t x = h.col;
      ++step;
      row[step] = x;
      e_prev = vb + idx;
      ec_prev = h.classes;
      s = v;
      v = x;
      if (step == walk_length) {
        valid_out[w.r] = 1;
        walking = false;
      } else {
        vb = hop_row(h);
        n = hop_deg(h);
        if (n == 0) {  // fugue.py:147
          for (int tt = step + 1; tt < L1; ++tt) row[tt] = -1;
          valid_out[w.r] = 0;
          walking = false;
        }
      }
    }
  }
}

}  // namespace n2v

// bytes of workspace the passes need for `total` walkers (two record lists + the counters)
int64_t n2v_walk_wedge2_workspace(int64_t total) {
  return (int64_t)n2v::kW2Counters * 4 + 2 * total * (int64_t)sizeof(n2v::WalkRec);
}

// returns 1 when the passes apply (and were enqueued), 0 when they do not, < 0 on error
int n2v_walk_wedge2_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                        int32_t num_walks, int32_t walk_length, double p, double q,
                        const n2v::UnitConsts &K, uint64_t seed, int32_t *walks_out,
                        uint8_t *valid_out, uint32_t *status, void *workspace,
                        int64_t workspace_bytes, int32_t rounds, void *stream) {
  using namespace n2v;
  if (!g->hops || !g->wedge_off || !g->wedge_pos || g->w || g->w64 || !K.dyadic) return 0;
  if (p == 1.0 && q == 1.0) return 0;
  if (g->reserved2 & N2V_HOPS_INLINE_RPOS) return 0;  // that hop table is the slots kernel's
  const int64_t total = n_start * (int64_t)num_walks;
  if (total >= 0xffffff00ll || walk_length >= (int32_t)kRecDone - 1) return 0;
  if (!workspace || workspace_bytes < n2v_walk_wedge2_workspace(total)) return 0;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return N2V_EINVAL;
  if (total == 0) return 1;
  if (rounds < 0) rounds = 0;
  if (rounds > 100) rounds = 100;
  hipStream_t st = (hipStream_t)stream;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(workspace);  // [k]: size of list k; [128 + k]: counters
  WalkRec *lists[2] = {reinterpret_cast<WalkRec *>(cnt + kW2Counters),
                       reinterpret_cast<WalkRec *>(cnt + kW2Counters) + total};
  if (hipMemsetAsync(cnt, 0, sizeof(uint32_t) * kW2Counters, st) != hipSuccess) return N2V_ELAUNCH;
  int64_t ib = (total + 255) / 256;
  if (ib > 65536) ib = 65536;
  hipLaunchKernelGGL(wedge2_init_kernel, dim3((unsigned)ib), dim3(256), 0, st, g->rowptr,
                     g->n_vertices, start_ids, total, num_walks, walk_length, lists[0], cnt, walks_out,
                     valid_out, status);
  // the return run shares a stack with "other" on ordinary rows: q > 1 with p > q, q < 1 with p < q
  const bool alone_under = K.bO <= 1.0 && K.bR >= K.bO, alone_over = K.bO >= 1.0 && K.bR <= K.bO;
  const int mode = alone_under ? 0 : alone_over ? 3 : 1;
#define N2V_W2_ROUNDS(M)                                                                           \
  do {                                                                                            \
    const int64_t cap = resident_blocks((const void *)wedge2_main_kernel<M>, kW2Threads, 0);      \
    int64_t rb = (total + kW2Threads * 4 - 1) / (kW2Threads * 4);                                 \
    const int64_t rcap = 8 * resident_blocks((const void *)wedge2_resolve_kernel<M>, kW2Threads, 0); \
    if (rb > rcap) rb = rcap;                                                                     \
    for (int k = 0; k < rounds; ++k) {                                                            \
      hipLaunchKernelGGL(wedge2_main_kernel<M>, dim3((unsigned)cap), dim3(kW2Threads), 0, st, *g, \
                         start_ids, num_walks, walk_length, q, K, seed, lists[k & 1], cnt + k,    \
                         cnt + 128 + k, walks_out, valid_out, status);                            \
      hipLaunchKernelGGL(wedge2_resolve_kernel<M>, dim3((unsigned)rb), dim3(kW2Threads), 0, st,   \
                         *g, start_ids, num_walks, walk_length, q, K, seed, lists[k & 1], cnt + k, \
                         lists[(k + 1) & 1], cnt + k + 1, walks_out, valid_out, status);          \
    }                                                                                             \
    const int64_t fcap = resident_blocks((const void *)wedge2_finish_kernel<M>, kW2Threads, 0);   \
    hipLaunchKernelGGL(wedge2_finish_kernel<M>, dim3((unsigned)fcap), dim3(kW2Threads), 0, st,    \
                       *g, start_ids, num_walks, walk_length, q, K, seed, lists[rounds & 1],      \
                       cnt + rounds, cnt + 128 + rounds, walks_out, valid_out, status);           \
  } while (0)
  if (mode == 0)
    N2V_W2_ROUNDS(0);
  else if (mode == 3)
    N2V_W2_ROUNDS(3);
  else
    N2V_W2_ROUNDS(1);
#undef N2V_W2_ROUNDS
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}
