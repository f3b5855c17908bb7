// Proposal-network losses for gfx950 (reference: /root/reference/nerfstudio/model_components/losses.py —
// outer :53-82, lossfun_outer :85-102, interlevel_loss :113-131, lossfun_distortion :135-146).
//
// Both are per-ray independent with tens to hundreds of samples: one wavefront per ray, rows staged in LDS, value
// and gradient in one pass (the losses are scalars whose upstream gradient is a known constant). The eager
// reference spends ~20 launches and several [N,S,S] temporaries on this; here it is two launches and O(N*S) bytes.
#include "ray_bodies.h"

namespace nsamd {


__global__ __launch_bounds__(kLossThreads) void interlevel_kernel(
    const float* __restrict__ c_in, const float* __restrict__ w_in, int Sf, const float* __restrict__ cp_in,
    const float* __restrict__ wp_in, int Sp, int64_t num_rays, float grad_scale, float* __restrict__ per_ray,
    float* __restrict__ dwp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  interlevel_body(lds + (size_t)wave_index() * interlevel_row_floats(Sf, Sp), c_in, w_in, Sf, cp_in, wp_in, Sp, num_rays,
                  grad_scale, per_ray, dwp);
}

__global__ __launch_bounds__(kLossThreads) void distortion_kernel(const float* __restrict__ s_bins,
                                                                  const float* __restrict__ weights, int S,
                                                                  int64_t num_rays, float grad_scale,
                                                                  float* __restrict__ per_ray,
                                                                  float* __restrict__ dw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  distortion_body(lds + (size_t)wave_index() * 2 * S, s_bins, weights, S, num_rays, grad_scale, per_ray, dw);
}

// All proposal losses of one training step in one launch: blockIdx.y < levels = interlevel loss of that proposal level,
// blockIdx.y == levels = distortion loss of the fine samples (models/nerfacto.py:367-375).
constexpr int kMaxPropLevels = 4;
struct PropLossArgs {
  const float* s_bins[kMaxPropLevels];
  const float* weights[kMaxPropLevels];
  float* per_ray[kMaxPropLevels];
  float* dw[kMaxPropLevels];
  int S[kMaxPropLevels];
  int levels;
};

__global__ __launch_bounds__(kLossThreads) void proposal_losses_kernel(
    const float* __restrict__ s_fine, const float* __restrict__ w_fine, int Sf, PropLossArgs a, int64_t num_rays,
    float inter_scale, float dist_scale, float* __restrict__ dist_per_ray, float* __restrict__ dw_dist) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int job = blockIdx.y;
  if (job < a.levels) {
    interlevel_body(lds + (size_t)wave_index() * interlevel_row_floats(Sf, a.S[job]), s_fine, w_fine, Sf, a.s_bins[job],
                    a.weights[job], a.S[job], num_rays, inter_scale, a.per_ray[job], a.dw[job]);
  } else {
    distortion_body(lds + (size_t)wave_index() * 2 * Sf, s_fine, w_fine, Sf, num_rays, dist_scale, dist_per_ray, dw_dist);
  }
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_proposal_losses(const float* s_bins_fine, const float* w_fine, int32_t S_fine, int32_t levels,
                                     const float* const* s_bins_prop, const float* const* w_prop,
                                     const int32_t* S_prop, int64_t num_rays, float interlevel_grad_scale,
                                     float distortion_grad_scale, float* const* interlevel_per_ray,
                                     float* const* dw_prop, float* distortion_per_ray, float* dw_distortion,
                                     nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S_fine > 0 && levels >= 0 && levels <= kMaxPropLevels);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(s_bins_fine && w_fine && distortion_per_ray);
  NSAMD_REQUIRE(levels == 0 || (s_bins_prop && w_prop && S_prop && interlevel_per_ray));
  if (S_fine > 2048) return NSAMD_ERR_UNSUPPORTED;
  PropLossArgs a{};
  a.levels = levels;
  size_t lds = sizeof(float) * 2 * S_fine * kLossRays;
  for (int i = 0; i < levels; ++i) {
    NSAMD_REQUIRE(s_bins_prop[i] && w_prop[i] && interlevel_per_ray[i] && S_prop[i] > 0);
    a.s_bins[i] = s_bins_prop[i];
    a.weights[i] = w_prop[i];
    a.per_ray[i] = interlevel_per_ray[i];
    a.dw[i] = dw_prop ? dw_prop[i] : nullptr;
    a.S[i] = S_prop[i];
    const size_t per_wave =
        sizeof(float) * ((2 * (S_fine + 2) + 2 * (S_prop[i] + 1) + (S_fine + 1) + 4 * S_fine + 1) & ~1);
    if (per_wave * kLossRays > lds) lds = per_wave * kLossRays;
  }
  if (lds > 64 * 1024) return NSAMD_ERR_UNSUPPORTED;
  dim3 g((unsigned)((num_rays + kLossRays - 1) / kLossRays), (unsigned)(levels + 1));
  proposal_losses_kernel<<<g, kLossThreads, lds, (hipStream_t)stream>>>(
      s_bins_fine, w_fine, S_fine, a, num_rays, interlevel_grad_scale, distortion_grad_scale, distortion_per_ray,
      dw_distortion);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_interlevel_loss(const float* s_bins_fine, const float* w_fine, int32_t S_fine,
                                     const float* s_bins_prop, const float* w_prop, int32_t S_prop,
                                     int64_t num_rays, float grad_scale, float* per_ray_loss, float* dw_prop,
                                     nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S_fine > 0 && S_prop > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(s_bins_fine && w_fine && s_bins_prop && w_prop && per_ray_loss);
  const size_t per_wave = sizeof(float) * ((2 * (S_fine + 2) + 2 * (S_prop + 1) + (S_fine + 1) + 4 * S_fine + 1) & ~1);
  if (per_wave * kLossRays > 64 * 1024) return NSAMD_ERR_UNSUPPORTED;
  const unsigned blocks = (unsigned)((num_rays + kLossRays - 1) / kLossRays);
  interlevel_kernel<<<blocks, kLossThreads, per_wave * kLossRays, (hipStream_t)stream>>>(
      s_bins_fine, w_fine, S_fine, s_bins_prop, w_prop, S_prop, num_rays, grad_scale, per_ray_loss, dw_prop);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_distortion_loss(const float* s_bins, const float* weights, int32_t S, int64_t num_rays,
                                     float grad_scale, float* per_ray_loss, float* dweights,
                                     nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(s_bins && weights && per_ray_loss);
  if (S > 2048) return NSAMD_ERR_UNSUPPORTED;
  const unsigned blocks = (unsigned)((num_rays + kLossRays - 1) / kLossRays);
  distortion_kernel<<<blocks, kLossThreads, sizeof(float) * 2 * S * kLossRays, (hipStream_t)stream>>>(
      s_bins, weights, S, num_rays, grad_scale, per_ray_loss, dweights);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
