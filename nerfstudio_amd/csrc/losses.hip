// Proposal-network losses for gfx950 (reference: /root/reference/nerfstudio/model_components/losses.py —
// outer :53-82, lossfun_outer :85-102, interlevel_loss :113-131, lossfun_distortion :135-146).
//
// Both are per-ray independent with tens to hundreds of samples: one wavefront per ray, rows staged in LDS, value
// and gradient in one pass (the losses are scalars whose upstream gradient is a known constant). The eager
// reference spends ~20 launches and several [N,S,S] temporaries on this; here it is two launches and O(N*S) bytes.
#include "common.h"
#include "wave.h"

namespace nsamd {

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

// LDS per wave: R[Sf+1] (double), cp[Sp+1], cy[Sp+1], c[Sf+1], w[Sf], r[Sf], lo[Sf], hi[Sf]
__device__ __forceinline__ void interlevel_body(
    float* lds, const float* __restrict__ c_in, const float* __restrict__ w_in, int Sf, const float* __restrict__ cp_in,
    const float* __restrict__ wp_in, int Sp, int64_t num_rays, float grad_scale, float* __restrict__ per_ray,
    float* __restrict__ dwp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kLossRays + wave;
  if (ray >= num_rays) return;
  const int per_wave = (2 * (Sf + 2) + 2 * (Sp + 1) + (Sf + 1) + 4 * Sf + 1) & ~1;  // floats, even: R stays 8-B aligned
  double* R = reinterpret_cast<double*>(lds + (size_t)wave * per_wave);
  float* cp = lds + (size_t)wave * per_wave + 2 * (Sf + 2);
  float* cy = cp + (Sp + 1);
  float* c = cy + (Sp + 1);
  float* w = c + (Sf + 1);
  float* rr = w + Sf;
  int* lo_i = reinterpret_cast<int*>(rr + Sf);
  int* hi_i = lo_i + Sf;
  for (int k = lane; k <= Sp; k += 64) cp[k] = cp_in[ray * (Sp + 1) + k];
  for (int i = lane; i <= Sf; i += 64) c[i] = c_in[ray * (Sf + 1) + i];
  for (int i = lane; i < Sf; i += 64) w[i] = w_in[ray * Sf + i];
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
__device__ __forceinline__ void distortion_body(float* lds, const float* __restrict__ s_bins,
                                                const float* __restrict__ weights, int S, int64_t num_rays,
                                                float grad_scale, float* __restrict__ per_ray,
                                                float* __restrict__ dw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kLossRays + wave;
  if (ray >= num_rays) return;
  float* mid = lds + wave * 2 * S;
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

__global__ __launch_bounds__(kLossThreads) void interlevel_kernel(
    const float* __restrict__ c_in, const float* __restrict__ w_in, int Sf, const float* __restrict__ cp_in,
    const float* __restrict__ wp_in, int Sp, int64_t num_rays, float grad_scale, float* __restrict__ per_ray,
    float* __restrict__ dwp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  interlevel_body(lds, c_in, w_in, Sf, cp_in, wp_in, Sp, num_rays, grad_scale, per_ray, dwp);
}

__global__ __launch_bounds__(kLossThreads) void distortion_kernel(const float* __restrict__ s_bins,
                                                                  const float* __restrict__ weights, int S,
                                                                  int64_t num_rays, float grad_scale,
                                                                  float* __restrict__ per_ray,
                                                                  float* __restrict__ dw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  distortion_body(lds, s_bins, weights, S, num_rays, grad_scale, per_ray, dw);
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
    interlevel_body(lds, s_fine, w_fine, Sf, a.s_bins[job], a.weights[job], a.S[job], num_rays, inter_scale,
                    a.per_ray[job], a.dw[job]);
  } else {
    distortion_body(lds, s_fine, w_fine, Sf, num_rays, dist_scale, dist_per_ray, dw_dist);
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
