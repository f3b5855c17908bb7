// Alpha compositing for gfx950 (reference: /root/reference/nerfstudio/model_components/renderers.py —
// RGBRenderer.combine_rgb :72-119 + eval nan_to_num/clamp :225-231, AccumulationRenderer :293-317,
// DepthRenderer median :354-364 / expected :365-383).
//
// One wavefront per ray: lane s owns sample s (stride 64 for S > 64), so the [S,3] rgb row and the weight row are
// read as contiguous, fully coalesced segments; the five running sums collapse with a 6-step xor-shuffle wave
// reduction. The median depth needs the reference's double-accumulated running weight sum (the integer index must
// match bit-for-bit): a wave scan in double. The expected depth is clipped to the batch-GLOBAL min/max of
// the sample midpoints as the reference does, which needs a device-wide min/max: every workgroup stores its partial
// min/max (no atomics: 8192 same-address device atomics cost ~100 us on this part, measured) and the finishing pass
// re-reduces the <= few thousand partials from L2 before clipping.
#include "ray_bodies.h"

namespace nsamd {

__global__ __launch_bounds__(kRenderThreads) void composite_fwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ weights, const float* __restrict__ t_bins,
    int64_t num_rays, int S, int background, float bg_r, float bg_g, float bg_b, int eval_mode,
    float* __restrict__ rgb_out, float* __restrict__ acc_out, float* __restrict__ depth_exp,
    float* __restrict__ depth_med, int32_t* __restrict__ med_idx, float* __restrict__ ws,
    const float* __restrict__ density, float* __restrict__ weights_out, const float* __restrict__ target,
    float grad_scale, float* __restrict__ sq_err, float* __restrict__ d_rgb_out, const float* __restrict__ bg_rays) {
  __shared__ float blk_min[kRaysPerBlock], blk_max[kRaysPerBlock];
  composite_fwd_body(blk_min, blk_max, rgb, weights, t_bins, num_rays, S, background, bg_r, bg_g, bg_b, eval_mode, rgb_out,
                     acc_out, depth_exp, depth_med, med_idx, ws, density, weights_out, target, grad_scale, sq_err, d_rgb_out,
                     bg_rays);
}


__global__ void depth_clip_kernel(float* __restrict__ depth, int64_t n, float* __restrict__ ws, int partials) {
  __shared__ float s_lo[256], s_hi[256];
  float lo = __uint_as_float(0x7f800000u), hi = __uint_as_float(0xff800000u);
  for (int i = threadIdx.x; i < partials; i += 256) {
    lo = fminf(lo, ws[2 + 2 * i]);
    hi = fmaxf(hi, ws[3 + 2 * i]);
  }
  s_lo[threadIdx.x] = lo;
  s_hi[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + o]);
      s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + o]);
    }
    __syncthreads();
  }
  lo = s_lo[0];
  hi = s_hi[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) { ws[0] = lo; ws[1] = hi; }  // kept for the backward's clip mask
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) depth[i] = fminf(fmaxf(depth[i], lo), hi);  // torch.clip(depth, steps.min(), steps.max())
}

__global__ __launch_bounds__(kRenderThreads) void composite_bwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ weights, const float* __restrict__ t_bins,
    int64_t num_rays, int S, int background, float bg_r, float bg_g, float bg_b,
    const float* __restrict__ d_rgb_out, const float* __restrict__ d_acc, const float* __restrict__ d_depth,
    const float* __restrict__ ws, const float* __restrict__ d_weights_add, float* __restrict__ d_rgb,
    float* __restrict__ d_weights, const float* __restrict__ density, float* __restrict__ d_density,
    const float* __restrict__ bg_rays) {
  extern __shared__ float lds[];
  composite_bwd_body(lds + (size_t)wave_index() * 3 * S, rgb, weights, t_bins, num_rays, S, background, bg_r, bg_g, bg_b,
                     d_rgb_out, d_acc, d_depth, ws, d_weights_add, d_rgb, d_weights, density, d_density, bg_rays);
}

int depth_clip_launch(float* depth, int64_t n, float* ws, int partials, hipStream_t stream) {
  depth_clip_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(depth, n, ws, partials);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

// MSELoss (model_components/losses.py:31 = nn.MSELoss, mean over all N*3 elements): value and gradient in one pass.
// loss_sum receives the SUM of squared errors (one atomic per workgroup; the caller zeroes it and divides by n).
__global__ void mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                float grad_scale, float* __restrict__ loss_sum, float* __restrict__ dpred) {
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - target[i];
    acc += d * d;
    if (dpred) dpred[i] = 2.0f * d * grad_scale;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) unsafeAtomicAdd(loss_sum, part[0] + part[1] + part[2] + part[3]);
}

// scale_gradients_by_distance_squared (model_components/losses.py:534-569, models/nerfacto.py:321-322): the forward is the
// identity; the gradients that reach the field's outputs are multiplied by clamp(((start + end) / 2)^2, 0, 1) per sample.
__global__ __launch_bounds__(256) void distance_gradient_scale_kernel(const float* __restrict__ t_bins, int64_t num_rays, int S,
                                                                      float* __restrict__ d_density,
                                                                      float* __restrict__ d_rgb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= num_rays * S) return;
  const int64_t ray = i / S;
  const int s_ = (int)(i - ray * S);
  const float* tb = t_bins + ray * (S + 1) + s_;
  const float mid = (tb[0] + tb[1]) / 2.0f;
  const float scale = fminf(fmaxf(mid * mid, 0.0f), 1.0f);
  if (d_density != nullptr) d_density[i] = d_density[i] * scale;
  if (d_rgb != nullptr) {
    d_rgb[3 * i] = d_rgb[3 * i] * scale;
    d_rgb[3 * i + 1] = d_rgb[3 * i + 1] * scale;
    d_rgb[3 * i + 2] = d_rgb[3 * i + 2] * scale;
  }
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_mse_loss(const float* pred, const float* target, int64_t n, float grad_scale, float* loss_sum,
                              float* dpred, nsamd_stream_t stream) {
  NSAMD_REQUIRE(n >= 0);
  if (n == 0) return NSAMD_OK;
  NSAMD_REQUIRE(pred && target);
  const unsigned blocks = (unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
  mse_loss_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(pred, target, n, grad_scale, loss_sum, dpred);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_composite_fwd(const float* rgb, const float* weights, const float* t_bins, int64_t num_rays,
                                   int32_t S, int background, const float* bg_rgb_host, int eval_mode,
                                   float* rgb_out, float* acc, float* depth_expected, float* depth_median,
                                   int32_t* median_idx, float* workspace, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(weights != nullptr);
  NSAMD_REQUIRE(background >= 0 && background <= 2);
  NSAMD_REQUIRE(rgb_out == nullptr || rgb != nullptr);
  NSAMD_REQUIRE(background != 2 || bg_rgb_host != nullptr);
  const bool need_t = depth_expected || depth_median || median_idx;
  NSAMD_REQUIRE(!need_t || t_bins != nullptr);
  NSAMD_REQUIRE(depth_expected == nullptr || workspace != nullptr);
  if (S > 4096) return NSAMD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  float* ws = workspace;
  const unsigned blocks = (unsigned)((num_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const float br = bg_rgb_host ? bg_rgb_host[0] : 0.f, bg = bg_rgb_host ? bg_rgb_host[1] : 0.f,
              bb = bg_rgb_host ? bg_rgb_host[2] : 0.f;
  composite_fwd_kernel<<<blocks, kRenderThreads, 0, st>>>(
      rgb, weights, need_t ? t_bins : nullptr, num_rays, S, background, br, bg, bb, eval_mode, rgb_out, acc,
      depth_expected, depth_median, median_idx, ws, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr, nullptr);
  NSAMD_CHECK_LAUNCH();
  if (depth_expected) {
    depth_clip_kernel<<<(unsigned)((num_rays + 255) / 256), 256, 0, st>>>(depth_expected, num_rays, ws, (int)blocks);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}

extern "C" int nsamd_composite_bwd(const float* rgb, const float* weights, const float* t_bins, int64_t num_rays,
                                   int32_t S, int background, const float* bg_rgb_host, const float* d_rgb_out,
                                   const float* d_acc, const float* d_depth, const float* workspace,
                                   const float* d_weights_add, float* d_rgb, float* d_weights,
                                   nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rgb && weights && d_rgb && d_weights);
  NSAMD_REQUIRE(background >= 0 && background <= 2);
  NSAMD_REQUIRE(background != 2 || bg_rgb_host != nullptr);
  NSAMD_REQUIRE(d_depth == nullptr || (t_bins != nullptr && workspace != nullptr));
  const unsigned blocks = (unsigned)((num_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const float br = bg_rgb_host ? bg_rgb_host[0] : 0.f, bg = bg_rgb_host ? bg_rgb_host[1] : 0.f,
              bb = bg_rgb_host ? bg_rgb_host[2] : 0.f;
  composite_bwd_kernel<<<blocks, kRenderThreads, 0, (hipStream_t)stream>>>(
      rgb, weights, t_bins, num_rays, S, background, br, bg, bb, d_rgb_out, d_acc, d_depth, workspace, d_weights_add, d_rgb,
      d_weights, nullptr, nullptr, nullptr);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_render_train(const float* rgb, const float* density, const float* t_bins, int64_t num_rays,
                                  int32_t S, int background, const float* bg_rgb_host, const float* target,
                                  float grad_scale, float* weights, float* rgb_out, float* acc, float* depth_expected,
                                  float* depth_median, float* workspace, float* sq_err, float* d_rgb_out,
                                  const float* bg_rays, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rgb && density && t_bins && weights && rgb_out);
  NSAMD_REQUIRE(background >= 0 && background <= 3);
  NSAMD_REQUIRE(background != 2 || bg_rgb_host != nullptr);
  NSAMD_REQUIRE(background != 3 || bg_rays != nullptr);
  NSAMD_REQUIRE(depth_expected == nullptr || workspace != nullptr);
  if (S > 4096) return NSAMD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((num_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const float br = bg_rgb_host ? bg_rgb_host[0] : 0.f, bg = bg_rgb_host ? bg_rgb_host[1] : 0.f,
              bb = bg_rgb_host ? bg_rgb_host[2] : 0.f;
  composite_fwd_kernel<<<blocks, kRenderThreads, 0, st>>>(rgb, nullptr, t_bins, num_rays, S, background, br, bg, bb, 0,
                                                          rgb_out, acc, depth_expected, depth_median, nullptr, workspace,
                                                          density, weights, target, grad_scale, sq_err, d_rgb_out, bg_rays);
  NSAMD_CHECK_LAUNCH();
  if (depth_expected) {
    depth_clip_kernel<<<(unsigned)((num_rays + 255) / 256), 256, 0, st>>>(depth_expected, num_rays, workspace,
                                                                          (int)blocks);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}

extern "C" int nsamd_render_train_bwd(const float* rgb, const float* weights, const float* density,
                                      const float* t_bins, int64_t num_rays, int32_t S, int background,
                                      const float* bg_rgb_host, const float* d_rgb_out, const float* d_weights_add,
                                      float* d_rgb, float* d_density, const float* bg_rays, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rgb && weights && density && t_bins && d_rgb_out && d_rgb && d_density);
  NSAMD_REQUIRE(background >= 0 && background <= 3);
  NSAMD_REQUIRE(background != 2 || bg_rgb_host != nullptr);
  NSAMD_REQUIRE(background != 3 || bg_rays != nullptr);
  if (S > 1024) return NSAMD_ERR_UNSUPPORTED;
  const unsigned blocks = (unsigned)((num_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const float br = bg_rgb_host ? bg_rgb_host[0] : 0.f, bg = bg_rgb_host ? bg_rgb_host[1] : 0.f,
              bb = bg_rgb_host ? bg_rgb_host[2] : 0.f;
  composite_bwd_kernel<<<blocks, kRenderThreads, sizeof(float) * 3 * kRaysPerBlock * (size_t)S, (hipStream_t)stream>>>(
      rgb, weights, t_bins, num_rays, S, background, br, bg, bb, d_rgb_out, nullptr, nullptr, nullptr, d_weights_add,
      d_rgb, nullptr, density, d_density, bg_rays);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_distance_gradient_scale(const float* t_bins, int64_t num_rays, int32_t S, float* d_density, float* d_rgb,
                                             nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0 || (d_density == nullptr && d_rgb == nullptr)) return NSAMD_OK;
  NSAMD_REQUIRE(t_bins != nullptr);
  const int64_t nb = (num_rays * S + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  distance_gradient_scale_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(t_bins, num_rays, S, d_density, d_rgb);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
