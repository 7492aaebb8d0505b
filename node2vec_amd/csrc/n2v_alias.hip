// n2v_alias.hip -- K1, first-order Walker alias tables for every CSR row (gfx950).
//
// For each row this is exactly generate_alias_tables(weights of the row)
// (reference randomwalk.py:157-190): probs = w / (sum(w) / n) in fp64, indices
// split into underfull (< 1.0) / overfull, LIFO pairing, leftovers keep alias 0.
// One wave64 per row: the split and the normalisation are lane-parallel, the
// pairing is replayed as two descending candidate streams exactly as in K2
// (n2v_walk.hip), here writing every slot instead of stopping at one.
// Output: packed 16-byte slots {col, alias vertex, prob}, CSR-aligned, so that one
// draw of the fast sampler is one 16-byte gather (the alias INDEX of the reference's
// table is resolved to the neighbour id it points to, row by row, at the end).  Algorithmic bytes: 16 V + 24 E
// (rowptr + col + w read, 16-byte slot written).
#include "n2v_alias_core.h"

namespace n2v {

__global__ __launch_bounds__(kWavesPerBlock * 64) void alias_build_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ w, const double *__restrict__ w64, int64_t n_rows,
    n2v_slot *__restrict__ slots, uint32_t *__restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  StepCtx c;
  c.need_cls = false;
  c.need_mem = false;
  c.p = 1.0;
  c.q = 1.0;
  c.s = -1;
  c.m = 1;
  c.iters = 1;
  for (int64_t rr = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block; rr < n_rows;
       rr += n_waves) {
    const int64_t row = readfirstlane_i64(rr);
    const int64_t vb = readfirstlane_i64(rowptr[row]);
    const int n = (int)(readfirstlane_i64(rowptr[row + 1]) - vb);
    if (n == 0) continue;
    c.vcol = col + vb;
    c.scol = col;
    c.vw = w ? w + vb : nullptr;
    c.vw64 = w64 ? w64 + vb : nullptr;
    c.n = n;
    c.nch = (n + 63) >> 6;
    n2v_slot *out = slots + vb;
    double unused;
    const double total = row_sum(c, lane, nullptr, -1, unused);
    const double avg = total / (double)n;
    if (avg == 0.0) {
      if (lane == 0) atomicOr(status, N2V_ST_ZERODIV);
      continue;
    }
    // alias = [0] * n ; probs = [x / avg for x in w]   (:171-173)
    for (int chunk = 0; chunk < c.nch; ++chunk) {
      bool valid;
      const double pr = chunk_bias<true>(c, chunk, lane, nullptr, valid) / avg;
      if (valid) {
        n2v_slot sl;
        sl.col = c.vcol[chunk * 64 + lane];
        sl.alias = 0;
        sl.prob = pr;
        out[chunk * 64 + lane] = sl;
      }
    }
    // the pairing below overwrites slots written above from another lane
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    int cu = c.nch, co = c.nch;
    uint64_t um = 0, om = 0;
    double uval = 0.0, oval = 0.0, dem_r = 0.0, r = 0.0;
    bool have_dem = false, have_o = false;
    int dem_idx = 0, o_idx = 0;
    for (;;) {  // :182-189
      double pu;
      int ui;
      if (have_dem) {
        pu = dem_r;
        ui = dem_idx;
        have_dem = false;
      } else {
        while (um == 0ull && cu > 0) {
          --cu;
          bool valid;
          uval = chunk_bias<true>(c, cu, lane, nullptr, valid) / avg;
          um = ballot64(valid && uval < 1.0);
        }
        if (um == 0ull) break;
        int l = 63 - __clzll((long long)um);
        um &= ~(1ull << l);
        pu = readlane_f64(uval, l);
        ui = cu * 64 + l;
      }
      if (!have_o) {
        while (om == 0ull && co > 0) {
          --co;
          bool valid;
          oval = chunk_bias<true>(c, co, lane, nullptr, valid) / avg;
          om = ballot64(valid && !(oval < 1.0));
        }
        if (om == 0ull) {  // `under` stays on its stack with its current value
          if (lane == 0) out[ui].prob = pu;
          break;
        }
        int l = 63 - __clzll((long long)om);
        om &= ~(1ull << l);
        r = readlane_f64(oval, l);
        o_idx = co * 64 + l;
        have_o = true;
      }
      if (lane == 0) {  // alias[under] = over; probs[under] is final
        out[ui].alias = o_idx;
        out[ui].prob = pu;
      }
      r = r + pu - 1.0;
      if (r < 1.0) {
        have_dem = true;
        dem_r = r;
        dem_idx = o_idx;
        have_o = false;
      }
    }
    if (have_o && lane == 0) out[o_idx].prob = r;  // overfull left on its stack
    // The table is complete with alias = index into the row (what the reference stores).
    // The walk sampler only ever needs the vertex behind that index, so resolve it here:
    // a draw is then ONE 16-byte gather {col, alias vertex, prob}.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    for (int chunk = 0; chunk < c.nch; ++chunk) {
      const int i = chunk * 64 + lane;
      if (i < n) out[i].alias = c.vcol[out[i].alias];
    }
  }
}

}  // namespace n2v

extern "C" int n2v_alias_build(const n2v_graph *g, n2v_slot *slots_out, uint32_t *status,
                               void *stream) {
  if (!g || !g->rowptr || !slots_out || !status || g->n_vertices < 0) return N2V_EINVAL;
  if (g->w && g->w64) return N2V_EINVAL;  // one storage form at most
  const int64_t n_rows = g->n_vertices;
  if (n_rows == 0 || g->n_edges == 0) return N2V_OK;
  if (!g->col) return N2V_EINVAL;
  int64_t blocks = (n_rows + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::alias_build_kernel,
                                           n2v::kWavesPerBlock * 64, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::alias_build_kernel, dim3((unsigned)blocks),
                     dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, g->rowptr, g->col,
                     g->w, g->w64, n_rows, slots_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
