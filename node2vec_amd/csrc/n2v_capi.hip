// n2v_capi.hip -- argument checking and dispatch for the C ABI of include/n2v_hip.h.
#include "n2v_common.h"

extern "C" {
int n2v_walk_exact_launch(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                          int32_t num_walks, int32_t walk_length, double p, double q,
                          uint64_t seed, int32_t *walks_out, uint8_t *valid_out,
                          uint32_t *status, void *stream);
int n2v_walk_exact_unit_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                            int32_t num_walks, int32_t walk_length, double p, double q,
                            uint64_t seed, int32_t *walks_out, uint8_t *valid_out,
                            uint32_t *status, void *workspace, int64_t workspace_bytes,
                            void *stream);
int64_t n2v_walk_exact_unit_workspace(const n2v_graph *g, int64_t total, int32_t walk_length,
                                      double p, double q);
int n2v_walk_uniform_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                         int32_t num_walks, int32_t walk_length, double p, double q,
                         uint64_t seed, int32_t *walks_out, uint8_t *valid_out,
                         uint32_t *status, void *stream);
int n2v_walk_fast_launch(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                         int32_t num_walks, int32_t walk_length, double p, double q,
                         uint64_t seed, int32_t *walks_out, uint8_t *valid_out,
                         uint32_t *status, void *stream);

int n2v_abi_version(void) { return N2V_ABI_VERSION; }

const char *n2v_status_string(int code) {
  switch (code) {
    case N2V_OK: return "ok";
    case N2V_EINVAL: return "invalid argument";
    case N2V_ELAUNCH: return "HIP launch/runtime error";
    case N2V_ENOGPU: return "no HIP device";
    default: return "unknown status";
  }
}

int n2v_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int64_t n2v_walk_workspace_bytes(const n2v_graph *g, int64_t n_start, int32_t num_walks,
                                 int32_t walk_length, double return_param, double inout_param,
                                 int32_t mode) {
  if (!g || n_start <= 0 || num_walks <= 0 || walk_length < 0 || mode != N2V_WALK_EXACT) return 0;
  if (return_param == 0.0 || inout_param == 0.0) return 0;
  return n2v_walk_exact_unit_workspace(g, n_start * (int64_t)num_walks, walk_length, return_param,
                                       inout_param);
}

int n2v_walk(const n2v_graph *g, const int32_t *start_ids, int64_t n_start, int32_t num_walks,
             int32_t walk_length, double return_param, double inout_param, uint64_t seed,
             int32_t mode, int32_t *walks_out, uint8_t *valid_out, uint32_t *status,
             void *stream) {
  return n2v_walk_ws(g, start_ids, n_start, num_walks, walk_length, return_param, inout_param, seed,
                     mode, walks_out, valid_out, status, nullptr, 0, stream);
}

int n2v_walk_ws(const n2v_graph *g, const int32_t *start_ids, int64_t n_start, int32_t num_walks,
                int32_t walk_length, double return_param, double inout_param, uint64_t seed,
                int32_t mode, int32_t *walks_out, uint8_t *valid_out, uint32_t *status,
                void *workspace, int64_t workspace_bytes, void *stream) {
  if (!g || !g->rowptr || !g->col || n_start < 0 || num_walks < 0 || walk_length < 0)
    return N2V_EINVAL;
  // buffers are only required when there is at least one walker
  const bool any = n_start > 0 && num_walks > 0;
  if (any && (!start_ids || !walks_out || !valid_out || !status)) return N2V_EINVAL;
  // generate_edge_alias_tables raises ValueError on p == 0 or q == 0 (randomwalk.py:214-217)
  if (return_param == 0.0 || inout_param == 0.0) return N2V_EINVAL;
  if (!any) return (mode == N2V_WALK_EXACT || mode == N2V_WALK_FAST) ? N2V_OK : N2V_EINVAL;
  if (g->w && g->w64) return N2V_EINVAL;  // one storage form at most
  if (mode == N2V_WALK_EXACT) {
    // every weight 1.0 (w == w64 == NULL): the specialised kernels -- p == q == 1 is two
    // gathers per step (n2v_walk_uniform.hip), other p, q n2v_walk_unit.hip
    const int ru = n2v_walk_uniform_try(g, start_ids, n_start, num_walks, walk_length,
                                        return_param, inout_param, seed, walks_out, valid_out,
                                        status, stream);
    if (ru != 0) return ru < 0 ? ru : N2V_OK;
    const int rc = n2v_walk_exact_unit_try(g, start_ids, n_start, num_walks, walk_length,
                                           return_param, inout_param, seed, walks_out,
                                           valid_out, status, workspace, workspace_bytes, stream);
    if (rc != 0) return rc < 0 ? rc : N2V_OK;
    // p == q == 1 (the reference's defaults, constants.py:22,26): w/p == w/q == w exactly, so
    // the table generate_edge_alias_tables builds at (s, v) IS generate_alias_tables(row v) --
    // the K1 slots when the caller has them.  One 16-byte gather per step instead of a
    // rebuild, same bits (the unbiased draw of the fast kernel is the exact draw).
    if (return_param == 1.0 && inout_param == 1.0 && g->slots)
      return n2v_walk_fast_launch(g, start_ids, n_start, num_walks, walk_length, return_param,
                                  inout_param, seed, walks_out, valid_out, status, stream);
    return n2v_walk_exact_launch(g, start_ids, n_start, num_walks, walk_length, return_param,
                                 inout_param, seed, walks_out, valid_out, status, stream);
  }
  if (mode == N2V_WALK_FAST) {
    // weighted graphs draw candidates from the K1 tables; unit-weight graphs from col itself
    if (!g->slots && (g->w || g->w64)) return N2V_EINVAL;
    return n2v_walk_fast_launch(g, start_ids, n_start, num_walks, walk_length, return_param,
                                inout_param, seed, walks_out, valid_out, status, stream);
  }
  return N2V_EINVAL;
}
}
