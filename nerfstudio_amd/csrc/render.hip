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
#include "common.h"
#include "wave.h"

namespace nsamd {

constexpr int kRenderThreads = 256;
constexpr int kRaysPerBlock = kRenderThreads / 64;

__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}


__global__ __launch_bounds__(kRenderThreads) void composite_fwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ weights, const float* __restrict__ t_bins,
    int64_t num_rays, int S, int background, float bg_r, float bg_g, float bg_b, int eval_mode,
    float* __restrict__ rgb_out, float* __restrict__ acc_out, float* __restrict__ depth_exp,
    float* __restrict__ depth_med, int32_t* __restrict__ med_idx, float* __restrict__ ws,
    const float* __restrict__ density, float* __restrict__ weights_out, const float* __restrict__ target,
    float grad_scale, float* __restrict__ sq_err, float* __restrict__ d_rgb_out, const float* __restrict__ bg_rays) {
  // background == 3 ("random", training): rgb_out is the composite WITHOUT a background (renderers.py:112-115) and the loss
  // is taken on rgb_out + bg_rays[ray] * (1 - acc) (blend_background_for_loss_computation, renderers.py:194-196).
  // density != nullptr (training step, nsamd_render_train): the weights are computed here from the densities
  // (RaySamples.get_weights, as sampler.hip) and written to weights_out; target != nullptr adds the per-ray squared
  // error and the MSE gradient of the composited colour.
  __shared__ float blk_min[kRaysPerBlock], blk_max[kRaysPerBlock];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  const bool want_minmax = t_bins != nullptr && depth_exp != nullptr;
  if (ray >= num_rays) {  // tail workgroup: idle waves still take part in the partial min/max
    if (want_minmax) {
      if (lane == 0) { blk_min[wave] = __uint_as_float(0x7f800000u); blk_max[wave] = __uint_as_float(0xff800000u); }
      __syncthreads();
    }
    return;
  }
  const float* w_in = density ? weights_out + ray * S : weights + ray * S;
  const float* tb = t_bins ? t_bins + ray * (S + 1) : nullptr;
  float sw = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f;
  float tmin = __uint_as_float(0x7f800000u), tmax = __uint_as_float(0xff800000u);
  double w_carry = 0.0;  // running sum of density * delta (double, as torch's CPU cumsum)
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    float w = 0.0f;
    if (density) {
      const float dd = s < S ? (tb[s + 1] - tb[s]) * density[ray * S + s] : 0.0f;
      double incl = (double)dd;
      incl = wave_scan_inclusive_f64(incl);
      incl = incl + w_carry;
      double excl = wave_shift_up1_f64(incl);
      if (lane == 0) excl = w_carry;
      w_carry = wave_read_f64<63>(incl);
      if (s < S) {
        w = nan_to_num((1.0f - expf(-dd)) * expf(-(float)excl));
        weights_out[ray * S + s] = w;
      }
    } else if (s < S) {
      w = w_in[s];
    }
    if (s >= S) continue;
    sw += w;
    if (rgb) {
      const float* c = rgb + (ray * S + s) * 3;
      float r = c[0], g = c[1], b = c[2];
      if (eval_mode) { r = nan_to_num(r); g = nan_to_num(g); b = nan_to_num(b); }
      sr += w * r;
      sg += w * g;
      sb += w * b;
    }
    if (tb) {
      const float step = (tb[s] + tb[s + 1]) / 2.0f;
      sd += w * step;
      tmin = fminf(tmin, step);
      tmax = fmaxf(tmax, step);
    }
  }
  sw = wave_sum(sw);
  if (rgb && rgb_out) {
    sr = wave_sum(sr);
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if (lane == 0) {
      float br = 0.f, bgc = 0.f, bb = 0.f;
      bool blend = false;
      if (background == 1) {  // "last_sample"  (renderers.py:112-114)
        const float* c = rgb + (ray * S + (S - 1)) * 3;
        br = c[0]; bgc = c[1]; bb = c[2];
        if (eval_mode) { br = nan_to_num(br); bgc = nan_to_num(bgc); bb = nan_to_num(bb); }
        blend = true;
      } else if (background == 2) {
        br = bg_r; bgc = bg_g; bb = bg_b;
        blend = true;
      }
      if (blend) {
        const float rem = 1.0f - sw;
        sr = sr + br * rem;
        sg = sg + bgc * rem;
        sb = sb + bb * rem;
      }
      if (eval_mode) {
        sr = fminf(fmaxf(sr, 0.f), 1.f);
        sg = fminf(fmaxf(sg, 0.f), 1.f);
        sb = fminf(fmaxf(sb, 0.f), 1.f);
      }
      rgb_out[ray * 3 + 0] = sr;
      rgb_out[ray * 3 + 1] = sg;
      rgb_out[ray * 3 + 2] = sb;
      if (target) {  // MSELoss value (per ray) and gradient, losses.py:31
        if (background == 3) {
          const float rem = 1.0f - sw;
          sr = sr + bg_rays[ray * 3 + 0] * rem;
          sg = sg + bg_rays[ray * 3 + 1] * rem;
          sb = sb + bg_rays[ray * 3 + 2] * rem;
        }
        const float dr = sr - target[ray * 3 + 0], dg = sg - target[ray * 3 + 1], db = sb - target[ray * 3 + 2];
        if (sq_err) sq_err[ray] = (dr * dr + dg * dg) + db * db;
        if (d_rgb_out) {
          d_rgb_out[ray * 3 + 0] = 2.0f * dr * grad_scale;
          d_rgb_out[ray * 3 + 1] = 2.0f * dg * grad_scale;
          d_rgb_out[ray * 3 + 2] = 2.0f * db * grad_scale;
        }
      }
    }
  }
  if (acc_out && lane == 0) acc_out[ray] = sw;
  if (tb && depth_exp) {
    sd = wave_sum(sd);
    tmin = wave_min(tmin);
    tmax = wave_max(tmax);
    if (lane == 0) {
      depth_exp[ray] = sd / (sw + 1e-10f);  // clipped by the finishing pass
      blk_min[wave] = tmin;
      blk_max[wave] = tmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float lo = blk_min[0], hi = blk_max[0];
#pragma unroll
      for (int i = 1; i < kRaysPerBlock; ++i) { lo = fminf(lo, blk_min[i]); hi = fmaxf(hi, blk_max[i]); }
      ws[2 + 2 * blockIdx.x] = lo;  // partials live after the two final words
      ws[3 + 2 * blockIdx.x] = hi;
    }
  }
  if (density) __threadfence_block();  // the median pass re-reads the weights this wave has just written
  if (tb && (depth_med || med_idx)) {
    // searchsorted(cumsum(w), 0.5, side="left"), clamped  (renderers.py:359-362). torch.cumsum (CPU) accumulates in
    // double and rounds each output to fp32: wave scan in double (see sampler.hip on why that is the same number).
    double carry = 0.0;
    int idx = S;
    for (int s0 = 0; s0 < S && idx == S; s0 += 64) {
      const int s2 = s0 + lane;
      double v = s2 < S ? (double)w_in[s2] : 0.0;
      v = wave_scan_inclusive_f64(v);
      v = v + carry;
      carry = wave_read_f64<63>(v);
      const unsigned long long hit = __ballot(s2 < S && (float)v >= 0.5f);
      if (hit != 0ull) idx = s0 + __builtin_ctzll(hit);
    }
    idx = min(idx, S - 1);
    if (lane == 0) {
      if (med_idx) med_idx[ray] = idx;
      if (depth_med) depth_med[ray] = (tb[idx] + tb[idx + 1]) / 2.0f;
    }
  }
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
  // density != nullptr (nsamd_render_train_bwd): d_weights is not stored; the gradient goes on through
  // RaySamples.get_weights to d_density (same formulas as weights_bwd_kernel in sampler.hip).
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  if (ray >= num_rays) return;
  const float* w_in = weights + ray * S;
  const float* tb = (d_depth && t_bins) ? t_bins + ray * (S + 1) : nullptr;
  float sw = 0.f, sd = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float w = w_in[s];
    sw += w;
    if (tb) sd += w * ((tb[s] + tb[s + 1]) / 2.0f);
  }
  sw = wave_sum(sw);
  sd = wave_sum(sd);
  const float gr = d_rgb_out ? d_rgb_out[ray * 3 + 0] : 0.f;
  const float gg = d_rgb_out ? d_rgb_out[ray * 3 + 1] : 0.f;
  const float gb = d_rgb_out ? d_rgb_out[ray * 3 + 2] : 0.f;
  const float ga = d_acc ? d_acc[ray] : 0.f;
  float br = 0.f, bgc = 0.f, bb = 0.f;
  if (background == 1) {
    const float* c = rgb + (ray * S + (S - 1)) * 3;
    br = c[0]; bgc = c[1]; bb = c[2];
  } else if (background == 2) {
    br = bg_r; bgc = bg_g; bb = bg_b;
  } else if (background == 3) {  // per-ray colour of the loss blend: d(pred + bg (1 - acc)) / d w = rgb - bg
    br = bg_rays[ray * 3 + 0]; bgc = bg_rays[ray * 3 + 1]; bb = bg_rays[ray * 3 + 2];
  }
  // expected depth = clip(num / (den + eps)); clip passes gradient inside [lo, hi] (inclusive)
  float g_num = 0.f, g_den = 0.f;
  if (tb) {
    const float den = sw + 1e-10f;
    const float raw = sd / den;
    const float lo = ws[0], hi = ws[1];
    const float gd = (raw >= lo && raw <= hi) ? d_depth[ray] : 0.f;
    g_num = gd / den;
    g_den = -gd * sd / (den * den);
  }
  const float bg_dot = gr * br + gg * bgc + gb * bb;  // d comp / d acc = -bg
  const float rem = 1.0f - sw;
  for (int s = lane; s < S; s += 64) {
    const float* c = rgb + (ray * S + s) * 3;
    const float w = w_in[s];
    float dw = gr * c[0] + gg * c[1] + gb * c[2] - bg_dot + ga + g_den;
    if (tb) dw += g_num * ((tb[s] + tb[s + 1]) / 2.0f);
    if (d_weights_add) dw += d_weights_add[ray * S + s];  // e.g. the distortion-loss gradient on the same weights
    if (density) lds[(size_t)wave * 3 * S + s] = dw;
    else d_weights[ray * S + s] = dw;
    float* o = d_rgb + (ray * S + s) * 3;
    float e = w;
    if (background == 1 && s == S - 1) e += rem;
    o[0] = gr * e;
    o[1] = gg * e;
    o[2] = gb * e;
  }
  if (density) {
    // d weights / d density: dd_j gets  gw_j T_j exp(-dd_j) - sum_{i>j} gw_i w_i, through delta_j
    const float* tbw = t_bins + ray * (S + 1);
    float* dwr = lds + (size_t)wave * 3 * S;
    float* ex_row = dwr + S;
    float* tr_row = ex_row + S;
    double carry = 0.0;
    for (int i0 = 0; i0 < S; i0 += 64) {
      const int i = i0 + lane;
      const float dd = i < S ? (tbw[i + 1] - tbw[i]) * density[ray * S + i] : 0.0f;
      double incl = (double)dd;
      incl = wave_scan_inclusive_f64(incl);
      incl = incl + carry;
      double excl = wave_shift_up1_f64(incl);
      if (lane == 0) excl = carry;
      carry = wave_read_f64<63>(incl);
      if (i < S) {
        ex_row[i] = expf(-dd);
        tr_row[i] = expf(-(float)excl);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    carry = 0.0;
    for (int r0 = 0; r0 < S; r0 += 64) {  // reversed order: exclusive suffix sums of gw * w
      const int r = r0 + lane;
      const int i = S - 1 - r;
      float ex = 0.f, trans = 0.f, g = 0.f;
      if (r < S) {
        ex = ex_row[i];
        trans = tr_row[i];
        const float w = (1.0f - ex) * trans;
        const bool finite = (w == w) && (fabsf(w) <= 3.4028234663852886e38f);
        g = finite ? dwr[i] : 0.0f;  // nan_to_num backward masks non-finite products
      }
      double incl = (double)(r < S ? g * ((1.0f - ex) * trans) : 0.0f);
      incl = wave_scan_inclusive_f64(incl);
      incl = incl + carry;
      double excl = wave_shift_up1_f64(incl);
      if (lane == 0) excl = carry;
      carry = wave_read_f64<63>(incl);
      if (r < S) d_density[ray * S + i] = (tbw[i + 1] - tbw[i]) * (g * trans * ex - (float)excl);
    }
  }
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
