// Ray samplers for gfx950 (reference: /root/reference/nerfstudio/model_components/ray_samplers.py,
// cameras/rays.py:129-152).
//
// Shape of the work: N rays x S (<= a few hundred) samples, a few MB per launch, all of it L2 resident. Two things
// matter: (1) every fp32 result that decides an integer sample index must equal the reference's torch-CPU value.
// ATen's CPU cumsum accumulates fp32 inputs left-to-right in DOUBLE and rounds every output to fp32
// (acc_type<float> = double), so the scans per ray (transmittance, weight sum, CDF) do exactly that on one lane,
// with IEEE div and no FMA contraction (this TU is built with -ffp-contract=off); (2) everything else is
// elementwise and uses all lanes.
// Layout: kRays rays per 256-thread workgroup; each ray's row is staged once in LDS with coalesced loads
// (row stride S+1 or S+2 floats = odd, so the one-lane-per-ray scan walks conflict-free banks), results leave
// through coalesced stores. N = 4096 gives 256 workgroups = one per CU.
#include "common.h"

namespace nsamd {

constexpr int kRays = 16;      // rays per workgroup
constexpr int kThreads = 256;  // 4 wavefronts

// ---------------------------------------------------------------------------------------------------------------
// UniformLinDispPiecewiseSampler (ray_samplers.py:78-128, 225-248): pure elementwise.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void piecewise_bins_kernel(const float* __restrict__ nears,
                                                                  const float* __restrict__ fars,
                                                                  const float* __restrict__ edges,
                                                                  const float* __restrict__ jitter,
                                                                  int64_t num_rays, int S, int spacing,
                                                                  float* __restrict__ s_bins,
                                                                  float* __restrict__ t_bins) {
  const int64_t total = num_rays * (int64_t)(S + 1);
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
    const int64_t ray = e / (S + 1);
    const int i = (int)(e - ray * (S + 1));
    float b = edges[i];
    if (jitter != nullptr) {
      // lower = [edges[0], centres], upper = [centres, edges[S]]   (ray_samplers.py:108-110)
      const float lower = (i == 0) ? edges[0] : (edges[i] + edges[i - 1]) / 2.0f;
      const float upper = (i == S) ? edges[S] : (edges[i + 1] + edges[i]) / 2.0f;
      b = lower + (upper - lower) * jitter[ray];
    }
    const float s_near = spacing_fn_mode(spacing, nears[ray]);
    const float s_far = spacing_fn_mode(spacing, fars[ray]);
    s_bins[e] = b;
    t_bins[e] = spacing_to_euclidean_mode(spacing, b, s_near, s_far);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// RaySamples.get_weights (cameras/rays.py:129-152)
// ---------------------------------------------------------------------------------------------------------------
// LDS: dd[kRays][S | 1] and acc[kRays][S | 1]  ("| 1" = row stride forced odd)
__global__ __launch_bounds__(kThreads) void weights_fwd_kernel(const float* __restrict__ t_bins,
                                                               const float* __restrict__ density,
                                                               int64_t num_rays, int S,
                                                               float* __restrict__ weights) {
  extern __shared__ float lds[];
  const int ld = S | 1;
  float* dd = lds;
  float* acc = lds + kRays * ld;
  const int64_t ray0 = (int64_t)blockIdx.x * kRays;
  const int nr = (int)min((int64_t)kRays, num_rays - ray0);
  for (int e = threadIdx.x; e < nr * S; e += kThreads) {
    const int r = e / S, i = e - r * S;
    const float* tb = t_bins + (ray0 + r) * (S + 1) + i;
    dd[r * ld + i] = (tb[1] - tb[0]) * density[(ray0 + r) * S + i];
  }
  __syncthreads();
  if (threadIdx.x < nr) {  // exclusive left-to-right cumsum, one lane per ray
    const float* d = dd + threadIdx.x * ld;
    float* a = acc + threadIdx.x * ld;
    double run = 0.0;  // torch.cumsum on CPU: double accumulator, fp32 outputs
#pragma unroll 8
    for (int i = 0; i < S; ++i) {
      a[i] = (float)run;
      run = run + (double)d[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nr * S; e += kThreads) {
    const int r = e / S, i = e - r * S;
    const float alpha = 1.0f - expf(-dd[r * ld + i]);
    const float trans = expf(-acc[r * ld + i]);
    weights[(ray0 + r) * S + i] = nan_to_num(alpha * trans);
  }
}

// d(weights)/d(density): dd_j gets  gw_j * T_j * exp(-dd_j)  -  sum_{i>j} gw_i * w_i
__global__ __launch_bounds__(kThreads) void weights_bwd_kernel(const float* __restrict__ t_bins,
                                                               const float* __restrict__ density,
                                                               const float* __restrict__ dweights,
                                                               int64_t num_rays, int S,
                                                               float* __restrict__ ddensity) {
  extern __shared__ float lds[];
  const int ld = S | 1;
  float* dd = lds;
  float* acc = lds + kRays * ld;     // exclusive cumsum
  float* gw = lds + 2 * kRays * ld;  // gw_i * w_i, then (in place) its exclusive suffix sums
  const int64_t ray0 = (int64_t)blockIdx.x * kRays;
  const int nr = (int)min((int64_t)kRays, num_rays - ray0);
  for (int e = threadIdx.x; e < nr * S; e += kThreads) {
    const int r = e / S, i = e - r * S;
    const float* tb = t_bins + (ray0 + r) * (S + 1) + i;
    dd[r * ld + i] = (tb[1] - tb[0]) * density[(ray0 + r) * S + i];
  }
  __syncthreads();
  if (threadIdx.x < nr) {
    const float* d = dd + threadIdx.x * ld;
    float* a = acc + threadIdx.x * ld;
    double run = 0.0;  // torch.cumsum on CPU: double accumulator, fp32 outputs
#pragma unroll 8
    for (int i = 0; i < S; ++i) {
      a[i] = (float)run;
      run = run + (double)d[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nr * S; e += kThreads) {
    const int r = e / S, i = e - r * S;
    const float ex = expf(-dd[r * ld + i]);
    const float trans = expf(-acc[r * ld + i]);
    const float w = (1.0f - ex) * trans;
    const bool finite = (w == w) && (fabsf(w) <= 3.4028234663852886e38f);
    // nan_to_num backward masks non-finite products
    gw[r * ld + i] = finite ? dweights[(ray0 + r) * S + i] * w : 0.0f;
  }
  __syncthreads();
  if (threadIdx.x < nr) {  // in-place exclusive suffix sums  suf_j = sum_{i>j} gw_i  (reverse cumsum, as autograd)
    float* q = gw + threadIdx.x * ld;
    double run = 0.0;
    for (int i = S - 1; i >= 0; --i) {
      const float v = q[i];
      q[i] = (float)run;
      run = run + (double)v;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nr * S; e += kThreads) {
    const int r = e / S, i = e - r * S;
    const float ex = expf(-dd[r * ld + i]);
    const float trans = expf(-acc[r * ld + i]);
    const float w = (1.0f - ex) * trans;
    const bool finite = (w == w) && (fabsf(w) <= 3.4028234663852886e38f);
    const float g = finite ? dweights[(ray0 + r) * S + i] : 0.0f;
    const float* tb = t_bins + (ray0 + r) * (S + 1) + i;
    ddensity[(ray0 + r) * S + i] = (tb[1] - tb[0]) * (g * trans * ex - gw[r * ld + i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PDFSampler.generate_ray_samples, include_original=False (ray_samplers.py:276-372)
// ---------------------------------------------------------------------------------------------------------------
// LDS: w[kRays][ldp], cdf[kRays][ldp] with ldp = (S_prev + 1) | 1, plus per-ray scalars.
__global__ __launch_bounds__(kThreads) void pdf_resample_kernel(
    const float* __restrict__ s_bins_prev, const float* __restrict__ weights, int S_prev,
    const float* __restrict__ u_base, const float* __restrict__ jitter, const float* __restrict__ nears,
    const float* __restrict__ fars, float anneal_host, const float* __restrict__ anneal_dev, float hist_pad, float eps,
    float u_offset, int spacing, int64_t num_rays, int S,
    float* __restrict__ s_bins, float* __restrict__ t_bins, int32_t* __restrict__ inds) {
  extern __shared__ float lds[];
  const int ldp = (S_prev + 1) | 1;
  float* w = lds;
  float* cdf = lds + kRays * ldp;
  float* wsum = lds + 2 * kRays * ldp;  // [kRays] padded sum
  float* wpad = wsum + kRays;           // [kRays] padding / S_prev
  const int64_t ray0 = (int64_t)blockIdx.x * kRays;
  const int nr = (int)min((int64_t)kRays, num_rays - ray0);
  const float anneal = anneal_dev ? anneal_dev[0] : anneal_host;  // device copy: graph-replayable schedules

  // (1) weights (annealed) + histogram padding                              ray_samplers.py:601, :303
  for (int e = threadIdx.x; e < nr * S_prev; e += kThreads) {
    const int r = e / S_prev, i = e - r * S_prev;
    float v = weights[(ray0 + r) * S_prev + i];
    if (anneal != 1.0f) v = powf(v, anneal);
    w[r * ldp + i] = v + hist_pad;
  }
  __syncthreads();
  // (2) left-to-right sum; padding for all-zero rays                         ray_samplers.py:306-309
  if (threadIdx.x < nr) {
    const float* q = w + threadIdx.x * ldp;
    double acc = 0.0;  // double-accumulated sum, rounded once (= cumsum(w)[-1] of the oracle; closest to torch.sum)
#pragma unroll 8
    for (int i = 0; i < S_prev; ++i) acc = acc + (double)q[i];
    const float run = (float)acc;
    const float pad = fmaxf(eps - run, 0.0f);
    wpad[threadIdx.x] = pad / (float)S_prev;
    wsum[threadIdx.x] = run + pad;
  }
  __syncthreads();
  // (3) pdf                                                                   ray_samplers.py:308-311
  for (int e = threadIdx.x; e < nr * S_prev; e += kThreads) {
    const int r = e / S_prev, i = e - r * S_prev;
    w[r * ldp + i] = (w[r * ldp + i] + wpad[r]) / wsum[r];
  }
  __syncthreads();
  // (4) cdf = [0, min(1, cumsum(pdf))]                                        ray_samplers.py:312-313
  if (threadIdx.x < nr) {
    const float* q = w + threadIdx.x * ldp;
    float* c = cdf + threadIdx.x * ldp;
    double run = 0.0;
    c[0] = 0.0f;
#pragma unroll 8
    for (int i = 0; i < S_prev; ++i) {
      run = run + (double)q[i];
      c[i + 1] = fminf(1.0f, (float)run);
    }
  }
  __syncthreads();
  // (5) inverse-CDF sampling of the S+1 new bin edges                         ray_samplers.py:315-358
  const int nb = S + 1;
  for (int e = threadIdx.x; e < nr * nb; e += kThreads) {
    const int r = e / nb, j = e - r * nb;
    const int64_t ray = ray0 + r;
    float u = u_base[j];
    if (jitter != nullptr) u = u + jitter[ray] / (float)nb;  // rand / num_bins   (ray_samplers.py:320)
    else u = u + u_offset;                                    // 1 / (2 num_bins)  (ray_samplers.py:327), host-rounded
    const float* c = cdf + r * ldp;
    // searchsorted(side="right"): number of cdf entries <= u
    int lo = 0, hi = S_prev + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (c[mid] <= u) lo = mid + 1;
      else hi = mid;
    }
    const int below = min(max(lo - 1, 0), S_prev);
    const int above = min(max(lo, 0), S_prev);
    const float c0 = c[below], c1 = c[above];
    const float* bp = s_bins_prev + ray * (S_prev + 1);
    const float b0 = bp[below], b1 = bp[above];
    float t = nan_to_num((u - c0) / (c1 - c0), 0.0f);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float b = b0 + t * (b1 - b0);
    const float s_near = spacing_fn_mode(spacing, nears[ray]);
    const float s_far = spacing_fn_mode(spacing, fars[ray]);
    s_bins[ray * nb + j] = b;
    t_bins[ray * nb + j] = spacing_to_euclidean_mode(spacing, b, s_near, s_far);
    if (inds != nullptr) inds[ray * nb + j] = lo;
  }
}

}  // namespace nsamd

using namespace nsamd;

static inline unsigned ray_blocks(int64_t num_rays) { return (unsigned)((num_rays + kRays - 1) / kRays); }

extern "C" int nsamd_piecewise_bins(const float* nears, const float* fars, const float* edges, const float* jitter,
                                    int64_t num_rays, int32_t S, int spacing, float* s_bins, float* t_bins,
                                    nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(nears && fars && edges && s_bins && t_bins);
  const int64_t total = num_rays * (int64_t)(S + 1);
  const unsigned blocks = (unsigned)min((int64_t)8192, (total + kThreads - 1) / kThreads);
  piecewise_bins_kernel<<<blocks, kThreads, 0, (hipStream_t)stream>>>(nears, fars, edges, jitter, num_rays, S, spacing,
                                                                      s_bins, t_bins);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_weights_fwd(const float* t_bins, const float* density, int64_t num_rays, int32_t S,
                                 float* weights, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_bins && density && weights);
  if (S > 1024) return NSAMD_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * 2 * kRays * (S | 1);
  weights_fwd_kernel<<<ray_blocks(num_rays), kThreads, lds, (hipStream_t)stream>>>(t_bins, density, num_rays, S,
                                                                                   weights);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_weights_bwd(const float* t_bins, const float* density, const float* dweights,
                                 int64_t num_rays, int32_t S, float* ddensity, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_bins && density && dweights && ddensity);
  if (S > 512) return NSAMD_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * 3 * kRays * (S | 1);
  weights_bwd_kernel<<<ray_blocks(num_rays), kThreads, lds, (hipStream_t)stream>>>(t_bins, density, dweights,
                                                                                   num_rays, S, ddensity);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_pdf_resample(const float* s_bins_prev, const float* weights, int32_t S_prev,
                                  const float* u_base, const float* jitter, const float* nears, const float* fars,
                                  float anneal, const float* anneal_dev, float histogram_padding, float eps,
                                  float u_offset, int spacing, int64_t num_rays, int32_t S, float* s_bins,
                                  float* t_bins, int32_t* inds, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0 && S_prev > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(s_bins_prev && weights && u_base && nears && fars && s_bins && t_bins);
  if (S_prev > 1024) return NSAMD_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * (2 * kRays * ((S_prev + 1) | 1) + 2 * kRays);
  pdf_resample_kernel<<<ray_blocks(num_rays), kThreads, lds, (hipStream_t)stream>>>(
      s_bins_prev, weights, S_prev, u_base, jitter, nears, fars, anneal, anneal_dev, histogram_padding, eps, u_offset,
      spacing, num_rays, S,
      s_bins, t_bins, inds);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
