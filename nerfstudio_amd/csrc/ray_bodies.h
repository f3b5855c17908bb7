// Per-ray device bodies (one wavefront per ray) shared by the stand-alone launches of render.hip / losses.hip / sampler.hip and
// by the training step's fused per-ray launch (fused_rays.hip: compositing + depth + MSE + proposal losses + compositing backward
// [+ the proposal levels' weights backward] in ONE launch). Each body is the code its kernel used to hold, moved here unchanged:
// the fused launch concatenates them inside one wave, so its results are the stand-alone launches' bits by construction.
// A body takes the wave's LDS row as a pointer (`lds_wave`); the caller lays the rows of a workgroup's waves out with ONE stride
// (the largest any body of the launch needs), so that waves of a workgroup in different bodies never overlap.
// Every body maps  ray = blockIdx.x * 4 + wave  (256-thread workgroups). Not part of the C ABI.
#pragma once

#include "common.h"
#include "wave.h"

namespace nsamd {

constexpr int kRenderThreads = 256;
constexpr int kRaysPerBlock = kRenderThreads / 64;

#if defined(__HIPCC__)
// Host (render.hip): the finishing pass of the expected depth — re-reduces the per-workgroup min / max partials in `ws` and clips
// depth[0..n) to the batch-global range of the sample midpoints (renderers.py:381). Returns an nsamd_status.
int depth_clip_launch(float* depth, int64_t n, float* ws, int partials, hipStream_t stream);
#endif

// ---- compositing (render.hip) ---------------------------------------------------------------------------------------


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


__device__ __forceinline__ void composite_fwd_body(float* blk_min, float* blk_max,
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
  const int lane = threadIdx.x & 63, wave = wave_index();
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

__device__ __forceinline__ void composite_bwd_body(float* lds_wave,
    const float* __restrict__ rgb, const float* __restrict__ weights, const float* __restrict__ t_bins,
    int64_t num_rays, int S, int background, float bg_r, float bg_g, float bg_b,
    const float* __restrict__ d_rgb_out, const float* __restrict__ d_acc, const float* __restrict__ d_depth,
    const float* __restrict__ ws, const float* __restrict__ d_weights_add, float* __restrict__ d_rgb,
    float* __restrict__ d_weights, const float* __restrict__ density, float* __restrict__ d_density,
    const float* __restrict__ bg_rays) {
  // density != nullptr (nsamd_render_train_bwd): d_weights is not stored; the gradient goes on through
  // RaySamples.get_weights to d_density (same formulas as weights_bwd_kernel in sampler.hip).
  const int lane = threadIdx.x & 63, wave = wave_index();
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
    if (density) lds_wave[s] = dw;
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
    float* dwr = lds_wave;
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

// ---- RaySamples.get_weights backward (sampler.hip) ----------------------------------------------------------------------
// d(weights)/d(density): dd_j gets  gw_j * T_j * exp(-dd_j)  -  sum_{i>j} gw_i * w_i
// LDS: per wave  ex[S], trans[S], gw[S]  (the reverse pass needs the forward values again)
__device__ __forceinline__ void weights_bwd_body(float* lds_wave, const float* __restrict__ t_bins,
                                                               const float* __restrict__ density,
                                                               const float* __restrict__ dweights,
                                                               int64_t num_rays, int S,
                                                               float* __restrict__ ddensity,
                                                               uint32_t* __restrict__ gate_out,
                                                               uint8_t* __restrict__ ray_mask) {
  const int lane = threadIdx.x & 63;
  const int wave = wave_index();
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  if (ray >= num_rays) return;  // wave-uniform; no workgroup barrier below
  float* ex_row = lds_wave;
  float* tr_row = ex_row + S;
  float* g_row = tr_row + S;
  const float* tb = t_bins + ray * (S + 1);
  const float* dn = density + ray * S;
  const float* dw = dweights + ray * S;
  {
    // A ray whose weights carry no gradient: with every optical thickness dt * density in [0, FLT_MAX] all forward
    // values are finite, so dL/d density = dt * (0 * T * e - 0) = dt * 0 for every sample — no scans needed. Anything
    // else (a non-zero or NaN upstream gradient, a NaN / Inf / negative thickness) keeps the full path: 0 * NaN must
    // stay NaN as in autograd. The interlevel loss reaches few rays (profiles/r02_study_proposal_sparsity.txt).
    // The first 256 samples' inputs in one burst of unconditional loads (clamped indices; all of a nerfacto level): in a loop
    // that loads where it tests — behind the short-circuit of `carries ||` — this pre-pass was one memory round trip per 64
    // samples, and it is all a ray without gradient does.
    float lo_pre[4], hi_pre[4], dn_pre[4], dw_pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = min(lane + 64 * q, S - 1);
      lo_pre[q] = tb[i];
      hi_pre[q] = tb[i + 1];
      dn_pre[q] = dn[i];
      dw_pre[q] = dw[i];
    }
    bool carries = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float dd = (hi_pre[q] - lo_pre[q]) * dn_pre[q];
      const bool c = dw_pre[q] != 0.0f || !(dd >= 0.0f && dd <= 3.4028234663852886e38f);
      carries = carries || (lane + 64 * q < S && c);
    }
    for (int i = lane + 256; i < S; i += 64) {
      const float dd = (tb[i + 1] - tb[i]) * dn[i];
      carries = carries || dw[i] != 0.0f || !(dd >= 0.0f && dd <= 3.4028234663852886e38f);
    }
    const bool ray_carries = __ballot(carries) != 0ull;
    if (ray_mask != nullptr && lane == 0) ray_mask[ray] = ray_carries ? 1 : 0;  // per-ray form of the flag (see nsamd.h)
    if (!ray_carries) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < S) ddensity[ray * S + lane + 64 * q] = (hi_pre[q] - lo_pre[q]) * 0.0f;
      for (int i = lane + 256; i < S; i += 64) ddensity[ray * S + i] = (tb[i + 1] - tb[i]) * 0.0f;
      return;
    }
    // some ray of this launch carries gradient: the rest of the level's backward chain has work to do. A PLAIN store of
    // the same value from every carrying wave (merged in the L2s, written back at the end of the kernel, which is what the
    // consumers — later kernels on the stream — need): write-through / atomic stores to ONE address are one fabric write
    // each (~88 per us chip-wide, MI355X_MICROARCH.md) — 4096 carrying rays cost tens of us that way (measured).
    if (gate_out != nullptr && lane == 0 && *reinterpret_cast<volatile uint32_t*>(gate_out) == 0u)
      *reinterpret_cast<volatile uint32_t*>(gate_out) = 1u;
  }
  double carry = 0.0;
  for (int i0 = 0; i0 < S; i0 += 64) {
    const int i = i0 + lane;
    const float dd = i < S ? (tb[i + 1] - tb[i]) * dn[i] : 0.0f;
    const double incl = carry + wave_scan_inclusive_f64((double)dd);
    double excl = wave_shift_up1_f64(incl);
    if (lane == 0) excl = carry;
    carry = wave_read_f64<63>(incl);
    if (i < S) {
      const float ex = expf(-dd);
      const float trans = expf(-(float)excl);
      const float w = (1.0f - ex) * trans;
      const bool finite = (w == w) && (fabsf(w) <= 3.4028234663852886e38f);
      const float g = finite ? dw[i] : 0.0f;  // nan_to_num backward masks non-finite products
      ex_row[i] = ex;
      tr_row[i] = trans;
      g_row[i] = g;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // exclusive suffix sums  suf_j = sum_{i>j} g_i w_i  (reverse cumsum, as autograd): scan the reversed row
  carry = 0.0;
  for (int r0 = 0; r0 < S; r0 += 64) {
    const int r = r0 + lane;      // reversed position
    const int i = S - 1 - r;      // element
    float ex = 0.0f, trans = 0.0f, g = 0.0f;
    if (r < S) { ex = ex_row[i]; trans = tr_row[i]; g = g_row[i]; }
    const float gw = g * ((1.0f - ex) * trans);
    const double incl = carry + wave_scan_inclusive_f64((double)(r < S ? gw : 0.0f));
    double excl = wave_shift_up1_f64(incl);
    if (lane == 0) excl = carry;
    carry = wave_read_f64<63>(incl);
    if (r < S) ddensity[ray * S + i] = (tb[i + 1] - tb[i]) * (g * trans * ex - (float)excl);
  }
}

// ---- UniformLinDispPiecewiseSampler (sampler.hip; ray_samplers.py:78-128, 225-248): the bin edges of one ray ------------------
__device__ __forceinline__ void piecewise_bins_body(int64_t ray, int lane, const float* __restrict__ nears,
                                                    const float* __restrict__ fars, const float* __restrict__ edges,
                                                    const float* __restrict__ jitter, int jitter_per_edge, int S, int spacing,
                                                    float* __restrict__ s_bins, float* __restrict__ t_bins) {
  const float s_near = spacing_fn_mode(spacing, nears[ray]);
  const float s_far = spacing_fn_mode(spacing, fars[ray]);
  // single_jitter: one draw per ray; otherwise one per bin edge, [num_rays, S+1] (ray_samplers.py:104-107)
  const float jit = (jitter != nullptr && !jitter_per_edge) ? jitter[ray] : 0.0f;
  float* sb = s_bins + ray * (S + 1);
  float* tb = t_bins + ray * (S + 1);
  for (int i = lane; i <= S; i += 64) {
    float b = edges[i];
    if (jitter != nullptr) {
      // lower = [edges[0], centres], upper = [centres, edges[S]]   (ray_samplers.py:108-110)
      const float lower = (i == 0) ? edges[0] : (edges[i] + edges[i - 1]) / 2.0f;
      const float upper = (i == S) ? edges[S] : (edges[i + 1] + edges[i]) / 2.0f;
      b = lower + (upper - lower) * (jitter_per_edge ? jitter[ray * (S + 1) + i] : jit);
    }
    sb[i] = b;
    tb[i] = spacing_to_euclidean_mode(spacing, b, s_near, s_far);
  }
}


// ---- PDFSampler.generate_ray_samples (sampler.hip; ray_samplers.py:276-372) ---------------------------------------------
// LDS: per wave  w[S_prev], cdf[S_prev + 1], the previous level's edges [S_prev + 1] (+ the S + 1 new edges when they are merged
// with the existing ones). kFused: also the level's RaySamples.get_weights and its median depth (see sampler.hip).
template <bool kFused>
__device__ __forceinline__ void pdf_resample_body(float* lds_wave, int64_t ray,
    const float* __restrict__ s_bins_prev, const float* __restrict__ weights, int S_prev,
    const float* __restrict__ u_base, const float* __restrict__ jitter, const float* __restrict__ nears,
    const float* __restrict__ fars, float anneal_host, const float* __restrict__ anneal_dev, float hist_pad, float eps,
    float u_offset, int spacing, int64_t num_rays, int S,
    float* __restrict__ s_bins, float* __restrict__ t_bins, int32_t* __restrict__ inds,
    const float* __restrict__ t_bins_prev, const float* __restrict__ density, float* __restrict__ weights_out,
    float* __restrict__ depth_median, int jitter_per_edge, int include_original) {
  const int lane = threadIdx.x & 63;
  float* w = lds_wave;
  float* cdf = w + S_prev;
  float* bprev = cdf + S_prev + 1;    // the previous level's spacing-domain edges (gathered by the search below)
  float* fresh = bprev + S_prev + 1;  // include_original only
  // Everything this ray reads from global memory is requested HERE, in one burst: the kernel is one wavefront per ray and
  // all rays are resident at once, so its duration is one wave's chain of latencies — loads issued where they are used
  // (behind the LDS fences) put five or six exposed round trips into it.
  const float anneal = anneal_dev ? anneal_dev[0] : anneal_host;  // device copy: graph-replayable schedules
  const int nb = S + 1;
  const float near_ray = nears[ray], far_ray = fars[ray];
  const float jit_ray = (jitter != nullptr && !jitter_per_edge) ? jitter[ray] : 0.0f;
  // (... and UNCONDITIONALLY, at clamped indices, with nothing consumed before the last one is out: a load under a lane
  //  predicate sits in a branch of its own, the compiler closes every such branch with s_waitcnt vmcnt(0), and the "burst" was
  //  four round trips in a row — read off the ISA)
  float u_pre[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) u_pre[c] = u_base[min(lane + 64 * c, nb - 1)];
  const float* bp = s_bins_prev + ray * (S_prev + 1);
  float bp_pre[5];  // edges 0 .. 319 of the previous level (all of them for the nerfacto counts; the rest below)
#pragma unroll
  for (int c = 0; c < 5; ++c) bp_pre[c] = bp[min(lane + 64 * c, S_prev)];
  float tb_lo[4] = {0.f, 0.f, 0.f, 0.f}, tb_hi[4] = {0.f, 0.f, 0.f, 0.f}, dn_pre[4] = {0.f, 0.f, 0.f, 0.f};
  if (kFused) {
    const float* tb0 = t_bins_prev + ray * (S_prev + 1);
    const float* dn0 = density + ray * S_prev;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = min(64 * c + lane, S_prev - 1);
      tb_lo[c] = tb0[i];
      tb_hi[c] = tb0[i + 1];
      dn_pre[c] = dn0[i];
    }
  }
#pragma unroll
  for (int c = 0; c < 5; ++c)
    if (lane + 64 * c <= S_prev) bprev[lane + 64 * c] = bp_pre[c];
  for (int i = lane + 320; i <= S_prev; i += 64) bprev[i] = bp[i];
  float dd_pre[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) dd_pre[c] = (64 * c + lane) < S_prev ? (tb_hi[c] - tb_lo[c]) * dn_pre[c] : 0.0f;

  if (kFused) {
    // (0) weights of the previous level  (cameras/rays.py:129-152), kept in the LDS row
    const float* tb = t_bins_prev + ray * (S_prev + 1);
    const float* dn = density + ray * S_prev;
    double carry0 = 0.0;
    for (int i0 = 0; i0 < S_prev; i0 += 64) {
      const int i = i0 + lane;
      const float dd = i0 == 0 ? dd_pre[0] : i0 == 64 ? dd_pre[1] : i0 == 128 ? dd_pre[2] : i0 == 192 ? dd_pre[3]
                       : (i < S_prev ? (tb[i + 1] - tb[i]) * dn[i] : 0.0f);  // (selects, not an indexed array: no scratch)
      const double incl = carry0 + wave_scan_inclusive_f64((double)dd);
      double excl = wave_shift_up1_f64(incl);
      if (lane == 0) excl = carry0;
      carry0 = wave_read_f64<63>(incl);
      if (i < S_prev) {
        const float alpha = 1.0f - expf(-dd);
        const float trans = expf(-(float)excl);
        const float wv = nan_to_num(alpha * trans);
        weights_out[ray * S_prev + i] = wv;
        w[i] = wv;
      }
    }
    if (depth_median != nullptr) {  // searchsorted(cumsum(w), 0.5, side="left"), clamped
      double carry1 = 0.0;
      int idx = S_prev;
      for (int i0 = 0; i0 < S_prev && idx == S_prev; i0 += 64) {
        const int i = i0 + lane;
        const double incl = carry1 + wave_scan_inclusive_f64(i < S_prev ? (double)w[i] : 0.0);
        carry1 = wave_read_f64<63>(incl);
        const unsigned long long hit = __ballot(i < S_prev && (float)incl >= 0.5f);
        if (hit != 0ull) idx = i0 + __builtin_ctzll(hit);
      }
      idx = min(idx, S_prev - 1);
      if (lane == 0) depth_median[ray] = (tb[idx] + tb[idx + 1]) / 2.0f;
    }
  }
  // (1) weights (annealed) + histogram padding, and their sum                 ray_samplers.py:601, :303-309
  double total = 0.0;
  for (int i0 = 0; i0 < S_prev; i0 += 64) {
    const int i = i0 + lane;
    float v = 0.0f;
    if (i < S_prev) {
      v = kFused ? w[i] : weights[ray * S_prev + i];
      // pow(weights, anneal) (ray_samplers.py:601): libm's powf. It is a quarter of this kernel (probe_sampler_clocks: 7.5 k
      // of 26.7 k clocks), and 2^(anneal log2 v) on v_log_f32 / v_exp_f32 brings the launch from 17.9 to 13.5 us — but the
      // PSNR stand-in then ends 0.5 dB lower on one of its three scenes in every twin run (profiles/r02_negative_results.txt),
      // so the accurate function stays.
      if (anneal != 1.0f) v = powf(v, anneal);
      v = v + hist_pad;
      w[i] = v;
    }
    total = total + wave_read_f64<63>(wave_scan_inclusive_f64((double)v));
  }
  const float run = (float)total;  // double-accumulated sum, rounded once (= cumsum(w)[-1] of the oracle)
  const float pad = fmaxf(eps - run, 0.0f);
  const float wpad = pad / (float)S_prev;
  const float wsum = run + pad;
  // (2) pdf and cdf = [0, min(1, cumsum(pdf))]                                 ray_samplers.py:308-313
  double carry = 0.0;
  if (lane == 0) cdf[0] = 0.0f;
  for (int i0 = 0; i0 < S_prev; i0 += 64) {
    const int i = i0 + lane;
    const float pdf = i < S_prev ? (w[i] + wpad) / wsum : 0.0f;
    const double incl = carry + wave_scan_inclusive_f64((double)pdf);
    carry = wave_read_f64<63>(incl);
    if (i < S_prev) cdf[i + 1] = fminf(1.0f, (float)incl);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // (3) inverse-CDF sampling of the S+1 new bin edges                         ray_samplers.py:315-358
  const float s_near = spacing_fn_mode(spacing, near_ray);
  const float s_far = spacing_fn_mode(spacing, far_ray);
  const int out_edges = include_original ? nb + S_prev + 1 : nb;
  for (int j = lane; j < nb; j += 64) {
    float u = j < 64 ? u_pre[0] : j < 128 ? u_pre[1] : u_base[j];
    // rand / num_bins: one draw per ray (single_jitter) or per new edge   (ray_samplers.py:318-322)
    if (jitter != nullptr) u = u + (jitter_per_edge ? jitter[ray * nb + j] : jit_ray) / (float)nb;
    else u = u + u_offset;                                    // 1 / (2 num_bins)  (ray_samplers.py:327), host-rounded
    // searchsorted(side="right"): number of cdf entries <= u
    int lo = 0, hi = S_prev + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1;
      else hi = mid;
    }
    const int below = min(max(lo - 1, 0), S_prev);
    const int above = min(max(lo, 0), S_prev);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bprev[below], b1 = bprev[above];
    float t = nan_to_num((u - c0) / (c1 - c0), 0.0f);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float b = b0 + t * (b1 - b0);
    if (include_original) {
      fresh[j] = b;
    } else {
      s_bins[ray * nb + j] = b;
      t_bins[ray * nb + j] = spacing_to_euclidean_mode(spacing, b, s_near, s_far);
    }
    if (inds != nullptr) inds[ray * nb + j] = lo;
  }
  if (include_original) {
    // sort(cat(existing, new)) (ray_samplers.py:356-357): both lists are ascending, so an element's place is its own index
    // plus the number of elements of the other list in front of it (existing edges first on ties — equal values either way)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* so = s_bins + ray * out_edges;
    float* to = t_bins + ray * out_edges;
    for (int i = lane; i <= S_prev; i += 64) {  // existing edge i: new edges strictly below it
      const float v = bprev[i];
      int lo = 0, hi = nb;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fresh[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      so[i + lo] = v;
      to[i + lo] = spacing_to_euclidean_mode(spacing, v, s_near, s_far);
    }
    for (int j = lane; j < nb; j += 64) {  // new edge j: existing edges at or below it
      const float v = fresh[j];
      int lo = 0, hi = S_prev + 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bprev[mid] <= v) lo = mid + 1;
        else hi = mid;
      }
      so[j + lo] = v;
      to[j + lo] = spacing_to_euclidean_mode(spacing, v, s_near, s_far);
    }
  }
}


// ---- proposal losses (losses.hip) -----------------------------------------------------------------------------------------

constexpr int kLossThreads = 256;
constexpr int kLossRays = kLossThreads / 64;
constexpr float kLossEps = 1.0e-7f;  // losses.py:35

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// number of entries of sorted a[0..n) that are <= v   (torch.searchsorted side="right")
__device__ __forceinline__ int upper_bound(const float* a, int n, float v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// first index in sorted int a[0..n) with a[i] >= v
__device__ __forceinline__ int lower_bound_i(const int* a, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
// number of entries of sorted int a[0..n) that are <= v
__device__ __forceinline__ int upper_bound_i(const int* a, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// floats of LDS per wave of interlevel_body (even: its double row stays 8-B aligned)
__host__ __device__ constexpr int interlevel_row_floats(int Sf, int Sp) {
  return (2 * (Sf + 2) + 2 * (Sp + 1) + (Sf + 1) + 4 * Sf + 1) & ~1;
}

// LDS per wave: R[Sf+1] (double), cp[Sp+1], cy[Sp+1], c[Sf+1], w[Sf], r[Sf], lo[Sf], hi[Sf]
__device__ __forceinline__ void interlevel_body(
    float* lds_wave, const float* __restrict__ c_in, const float* __restrict__ w_in, int Sf, const float* __restrict__ cp_in,
    const float* __restrict__ wp_in, int Sp, int64_t num_rays, float grad_scale, float* __restrict__ per_ray,
    float* __restrict__ dwp, const float* __restrict__ dens_fine = nullptr,
    const float* __restrict__ t_fine = nullptr) {
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  if (ray >= num_rays) return;
  double* R = reinterpret_cast<double*>(lds_wave);
  float* cp = lds_wave + 2 * (Sf + 2);
  float* cy = cp + (Sp + 1);
  float* c = cy + (Sp + 1);
  float* w = c + (Sf + 1);
  float* rr = w + Sf;
  int* lo_i = reinterpret_cast<int*>(rr + Sf);
  int* hi_i = lo_i + Sf;
  for (int k = lane; k <= Sp; k += 64) cp[k] = cp_in[ray * (Sp + 1) + k];
  for (int i = lane; i <= Sf; i += 64) c[i] = c_in[ray * (Sf + 1) + i];
  if (dens_fine != nullptr) {
    // the fine level's weights from its densities (RaySamples.get_weights, cameras/rays.py:129-152): the same operations as
    // composite_fwd_body, so the same bits — a launch that composites and takes the losses at once (round 5's merged launch: csrc/experiments)
    // has no weight row in memory yet when this wave starts
    const float* tbf = t_fine + ray * (Sf + 1);
    double w_carry = 0.0;
    for (int s0 = 0; s0 < Sf; s0 += 64) {
      const int s = s0 + lane;
      const float dd = s < Sf ? (tbf[s + 1] - tbf[s]) * dens_fine[ray * Sf + s] : 0.0f;
      double incl = (double)dd;
      incl = wave_scan_inclusive_f64(incl);
      incl = incl + w_carry;
      double excl = wave_shift_up1_f64(incl);
      if (lane == 0) excl = w_carry;
      w_carry = wave_read_f64<63>(incl);
      if (s < Sf) w[s] = nan_to_num((1.0f - expf(-dd)) * expf(-(float)excl));
    }
  } else {
    for (int i = lane; i < Sf; i += 64) w[i] = w_in[ray * Sf + i];
  }
  {  // cy = [0, cumsum(wp)]   (losses.py:69); double-accumulated like torch's CPU cumsum, as a wave scan
    double carry = 0.0;
    if (lane == 0) cy[0] = 0.0f;
    for (int k0 = 0; k0 < Sp; k0 += 64) {
      const int k = k0 + lane;
      double v = k < Sp ? (double)wp_in[ray * Sp + k] : 0.0;
      v = wave_scan_inclusive_f64(v);
      v = v + carry;
      carry = wave_read_f64<63>(v);
      if (k < Sp) cy[k + 1] = (float)v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float loss = 0.0f;
  for (int i = lane; i < Sf; i += 64) {
    int lo = upper_bound(cp, Sp, c[i]) - 1;          // starts = cp[0..Sp)      (losses.py:71-72)
    lo = min(max(lo, 0), Sp - 1);
    int hi = upper_bound(cp + 1, Sp, c[i + 1]);      // ends   = cp[1..Sp]      (losses.py:73-74)
    hi = min(max(hi, 0), Sp - 1);
    const float outer = cy[hi + 1] - cy[lo];
    const float diff = w[i] - outer;
    const float clipped = fmaxf(diff, 0.0f);
    loss += clipped * clipped / (w[i] + kLossEps);
    rr[i] = 2.0f * clipped / (w[i] + kLossEps);      // = - d loss_i / d outer_i
    lo_i[i] = lo;
    hi_i[i] = hi;
  }
  loss = wave_sum_l(loss);
  if (lane == 0) per_ray[ray] = loss;
  if (dwp != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // d loss / d wp_k = - sum over the fine intervals i whose [lo_i, hi_i] covers k of rr_i. With sorted bin edges
    // lo and hi are non-decreasing in i, so the cover of k is one contiguous range of i: two binary searches and a
    // difference of (double) prefix sums replace the O(Sf) loop per k. Unsorted input keeps the direct loop.
    bool sorted_ok = true;
    for (int i = lane; i < Sf; i += 64)
      if (i > 0 && (lo_i[i] < lo_i[i - 1] || hi_i[i] < hi_i[i - 1])) sorted_ok = false;
    if (__ballot(!sorted_ok) == 0ull) {
      double carry = 0.0;
      if (lane == 0) R[0] = 0.0;
      for (int i0 = 0; i0 < Sf; i0 += 64) {
        const int i = i0 + lane;
        double v = i < Sf ? (double)rr[i] : 0.0;
        v = wave_scan_inclusive_f64(v);
        v = v + carry;
        carry = wave_read_f64<63>(v);
        if (i < Sf) R[i + 1] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < Sp; k += 64) {
        const int first = lower_bound_i(hi_i, Sf, k);     // first i with hi_i >= k
        const int last = upper_bound_i(lo_i, Sf, k) - 1;  // last i with lo_i <= k
        const float g = first <= last ? -(float)(R[last + 1] - R[first]) : 0.0f;
        dwp[ray * Sp + k] = g * grad_scale;
      }
    } else {
      for (int k = lane; k < Sp; k += 64) {
        float g = 0.0f;
        for (int i = 0; i < Sf; ++i) g -= (lo_i[i] <= k && k <= hi_i[i]) ? rr[i] : 0.0f;
        dwp[ray * Sp + k] = g * grad_scale;
      }
    }
  }
}

// LDS per wave: mid[S], w[S]
__device__ __forceinline__ void distortion_body(float* lds_wave, const float* __restrict__ s_bins,
                                                const float* __restrict__ weights, int S, int64_t num_rays,
                                                float grad_scale, float* __restrict__ per_ray,
                                                float* __restrict__ dw) {
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  if (ray >= num_rays) return;
  float* mid = lds_wave;
  float* w = mid + S;
  const float* b = s_bins + ray * (S + 1);
  for (int i = lane; i < S; i += 64) {
    mid[i] = (b[i + 1] + b[i]) / 2.0f;
    w[i] = weights[ray * S + i];
  }
  __builtin_amdgcn_wave_barrier();
  float loss = 0.0f;
  for (int i = lane; i < S; i += 64) {
    const float mi = mid[i], wi = w[i];
    float inner = 0.0f;
    for (int k = 0; k < S; ++k) inner += w[k] * fabsf(mi - mid[k]);
    const float delta = b[i + 1] - b[i];
    loss += wi * inner + wi * wi * delta / 3.0f;
    if (dw != nullptr) dw[ray * S + i] = (2.0f * inner + 2.0f * wi * delta / 3.0f) * grad_scale;
  }
  loss = wave_sum_l(loss);
  if (lane == 0) per_ray[ray] = loss;
}

}  // namespace nsamd
