// Backward of ALL proposal levels of an iteration that updates the proposal networks (ProposalNetworkSampler,
// /root/reference/nerfstudio/model_components/ray_samplers.py:590-609 -> RaySamples.get_weights backward, cameras/rays.py:129-152
// -> HashMLPDensityField backward, fields/density_fields.py:94-117 -> HashEncoding backward, field_components/encodings.py:417-458)
// as one entry point: nsamd_proposal_levels_bwd.
//
// Per level the chain is six launches (weights backward with the zero-gradient gate, density-MLP backward, the fixed-order
// reduce of its weight-gradient partials, the scatter's route / apply / finish passes), and every one of them is as long as
// its slowest workgroup's chain of memory round trips, not as its work (profiles/r06_s25_*, r06_s26_*: the run kernel's live
// waves spend 16 - 30 us in two sweeps whether 5 or 900 of 4096 rays carry gradient). The levels share nothing — separate
// networks, tables, gradients, scratch —, so here the SAME stage of two levels is one launch (blockIdx.y / .z selects the
// level, every workgroup runs the unchanged per-level body): twelve launches become six, and two latency chains run side
// by side. Same bits as the per-level entry points, call by call.
#include "proposal_chain.h"
#include "scatter.h"

using namespace nsamd;

static int level_checks(const nsamd_proposal_level_bwd& l) {
  NSAMD_REQUIRE(l.num_rays > 0 && l.samples_per_ray > 0 && l.samples_per_ray <= 1024);
  NSAMD_REQUIRE(l.t_bins && l.density && l.dweights && l.ddensity && l.gate && l.ray_mask);
  NSAMD_REQUIRE(l.enc && l.pre && l.denc && l.dW0 && l.db0 && l.dW1 && l.db1 && l.mlp.W0 && l.mlp.b0 && l.mlp.W1 && l.mlp.b1);
  NSAMD_REQUIRE(l.origins && l.directions && l.table && l.dtable && l.scatter_workspace);
  NSAMD_REQUIRE(l.transform >= 0 && l.transform <= 2);
  NSAMD_REQUIRE(l.grid.num_levels > 0 && l.grid.num_levels <= NSAMD_MAX_LEVELS && 2 * l.grid.num_levels == l.mlp.in_dim);
  if (l.grid.log2_table_size < 1 || l.grid.log2_table_size > 28) return NSAMD_ERR_UNSUPPORTED;
  return NSAMD_OK;
}

static nsamd_points level_points(const nsamd_proposal_level_bwd& l) {
  nsamd_points p{};
  p.positions = nullptr;
  p.origins = l.origins;
  p.directions = l.directions;
  p.t_bins = l.t_bins;
  p.samples_per_ray = l.samples_per_ray;
  return p;
}

// one level through the per-level entry points (an odd level out, or a stage whose two calls cannot share a launch)
static int weights_single(const nsamd_proposal_level_bwd& l, nsamd_stream_t st) {
  return nsamd_weights_bwd_gate(l.t_bins, l.density, l.dweights, l.num_rays, l.samples_per_ray, l.ddensity, l.gate, l.ray_mask,
                                /*gate_precleared=*/1, st);
}
static int density_single(const nsamd_proposal_level_bwd& l, nsamd_stream_t st) {
  return nsamd_density_mlp_bwd_gated(l.enc, l.selector, l.pre, l.ddensity, l.num_rays * l.samples_per_ray, l.mlp, l.denc, l.dW0,
                                     l.db0, l.dW1, l.db1, l.mlp_workspace, l.mlp_workspace_floats, l.gate, l.ray_mask,
                                     l.samples_per_ray, st);
}
static int scatter_single(const nsamd_proposal_level_bwd& l, nsamd_stream_t st) {
  const int64_t M = l.num_rays * l.samples_per_ray;
  return nsamd_hashgrid_encode_bwd_gated(level_points(l), M, l.transform, l.aabb, l.table, l.grid, l.denc, 1, M, l.dtable,
                                         l.scatter_workspace, l.scatter_workspace_floats, l.gate, l.ray_mask, st);
}

extern "C" int nsamd_proposal_levels_bwd(const nsamd_proposal_level_bwd* levels, int32_t num_levels, int32_t gates_precleared,
                                         nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_levels >= 0 && (num_levels == 0 || levels != nullptr));
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < num_levels; ++i) {
    const int rc = level_checks(levels[i]);
    if (rc) return rc;
  }
  if (!gates_precleared) {
    for (int i = 0; i < num_levels; ++i)
      if (hipMemsetAsync(levels[i].gate, 0, sizeof(uint32_t), st) != hipSuccess) return NSAMD_ERR_LAUNCH;
  }
  for (int i = 0; i < num_levels; i += 2) {
    const nsamd_proposal_level_bwd& a = levels[i];
    if (i + 1 >= num_levels) {
      int rc = weights_single(a, stream);
      if (!rc) rc = density_single(a, stream);
      if (!rc) rc = scatter_single(a, stream);
      if (rc) return rc;
      break;
    }
    const nsamd_proposal_level_bwd& b = levels[i + 1];
    const int64_t Ma = a.num_rays * a.samples_per_ray, Mb = b.num_rays * b.samples_per_ray;
    // ---- RaySamples.get_weights backward + gate ------------------------------------------------------------------------
    int rc = weights_bwd_launch_pair(
        WeightsBwdCall{a.t_bins, a.density, a.dweights, a.num_rays, a.samples_per_ray, a.ddensity, a.gate, a.ray_mask},
        WeightsBwdCall{b.t_bins, b.density, b.dweights, b.num_rays, b.samples_per_ray, b.ddensity, b.gate, b.ray_mask}, st);
    if (rc == NSAMD_ERR_UNSUPPORTED) {
      rc = weights_single(a, stream);
      if (!rc) rc = weights_single(b, stream);
    }
    if (rc) return rc;
    // ---- density MLP backward + weight-gradient reduce -------------------------------------------------------------------
    rc = density_bwd_launch_pair(
        DensityBwdCall{a.enc, a.selector, a.pre, a.ddensity, Ma, a.mlp, a.denc, a.dW0, a.db0, a.dW1, a.db1, a.mlp_workspace,
                       a.mlp_workspace_floats, a.gate, a.ray_mask, a.samples_per_ray},
        DensityBwdCall{b.enc, b.selector, b.pre, b.ddensity, Mb, b.mlp, b.denc, b.dW0, b.db0, b.dW1, b.db1, b.mlp_workspace,
                       b.mlp_workspace_floats, b.gate, b.ray_mask, b.samples_per_ray},
        st);
    if (rc == NSAMD_ERR_UNSUPPORTED) {
      rc = density_single(a, stream);
      if (!rc) rc = density_single(b, stream);
    }
    if (rc) return rc;
    // ---- table scatter: route, apply, finish -----------------------------------------------------------------------------
    ScatterPlan pa = scatter_plan(a.grid, Ma, false), pb = scatter_plan(b.grid, Mb, false);
    rc = NSAMD_ERR_UNSUPPORTED;
    if (pa.ok && pb.ok && pa.total_words <= a.scatter_workspace_floats && pb.total_words <= b.scatter_workspace_floats) {
      rc = scatter_launch_pair(
          ScatterCall{level_points(a), Ma, a.transform, a.aabb, a.grid, a.denc, 1, Ma, a.dtable, a.scatter_workspace, pa, false,
                      a.gate, a.ray_mask},
          ScatterCall{level_points(b), Mb, b.transform, b.aabb, b.grid, b.denc, 1, Mb, b.dtable, b.scatter_workspace, pb, false,
                      b.gate, b.ray_mask},
          st);
    }
    if (rc == NSAMD_ERR_UNSUPPORTED) {
      rc = scatter_single(a, stream);
      if (!rc) rc = scatter_single(b, stream);
    }
    if (rc) return rc;
  }
  return NSAMD_OK;
}
