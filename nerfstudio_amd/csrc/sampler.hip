// Ray samplers for gfx950 (reference: /root/reference/nerfstudio/model_components/ray_samplers.py,
// cameras/rays.py:129-152).
//
// Shape of the work: N rays x S (<= a few hundred) samples, a few MB per launch, all of it L2 resident. Two things
// matter: (1) every fp32 result that decides an integer sample index must equal the reference's torch-CPU value.
// ATen's CPU cumsum accumulates fp32 inputs left-to-right in DOUBLE and rounds every output to fp32
// (acc_type<float> = double), so the scans per ray (transmittance, weight sum, CDF) accumulate in double too,
// with IEEE div and no FMA contraction (this TU is built with -ffp-contract=off); (2) the launch is tiny, so
// latency decides: one wavefront per ray, wave-level scans, no workgroup barriers.
// Layout: 4 rays per 256-thread workgroup (N = 4096 -> 1024 workgroups); rows are read/written coalesced.
#include <stdlib.h>

#include "ray_bodies.h"

NSAMD_PROBE_DEFINE(sampler)

namespace nsamd {

constexpr int kThreads = 256;  // 4 wavefronts

// ---------------------------------------------------------------------------------------------------------------
// UniformLinDispPiecewiseSampler (ray_samplers.py:78-128, 225-248): pure elementwise.
// ---------------------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(kThreads) void piecewise_bins_kernel(const float* __restrict__ nears,
                                                                  const float* __restrict__ fars,
                                                                  const float* __restrict__ edges,
                                                                  const float* __restrict__ jitter,
                                                                  int jitter_per_edge, int64_t num_rays, int S,
                                                                  int spacing, float* __restrict__ s_bins,
                                                                  float* __restrict__ t_bins) {
  // one wavefront per ray (4 rays per workgroup): per-ray scalars are computed once, the edge index is a 32-bit loop
  // counter (the flat-index version spent its time in 64-bit divisions)
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  piecewise_bins_body(ray, lane, nears, fars, edges, jitter, jitter_per_edge, S, spacing, s_bins, t_bins);
}

// nsamd_select_batch + nsamd_piecewise_bins in one launch (the first two launches of a training iteration over a pool of ray
// batches; neither depends on the other — the bins need nears / fars / the jitter draw, not the rays): the ray's wave copies
// its origin, direction, target colour and camera index out of the pool slot, then writes its initial bins.
__global__ __launch_bounds__(kThreads) void select_bins_kernel(
    const float* __restrict__ slot_dev, int32_t slots, int64_t num_rays, const float* __restrict__ origins_pool,
    const float* __restrict__ directions_pool, const int64_t* __restrict__ cameras_pool, const float* __restrict__ target_pool,
    float* __restrict__ origins, float* __restrict__ directions, int64_t* __restrict__ cameras, float* __restrict__ target,
    const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ edges,
    const float* __restrict__ jitter, int jitter_per_edge, int S, int spacing, float* __restrict__ s_bins,
    float* __restrict__ t_bins) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  int32_t slot = (int32_t)slot_dev[0];
  slot = slot < 0 ? 0 : (slot >= slots ? slots - 1 : slot);
  if (lane < 3) {
    const int64_t dst = 3 * ray + lane, src = (int64_t)slot * 3 * num_rays + dst;
    origins[dst] = origins_pool[src];
    directions[dst] = directions_pool[src];
    target[dst] = target_pool[src];
  } else if (lane == 3) {
    cameras[ray] = cameras_pool[(int64_t)slot * num_rays + ray];
  }
  piecewise_bins_body(ray, lane, nears, fars, edges, jitter, jitter_per_edge, S, spacing, s_bins, t_bins);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-ray scans: one WAVEFRONT per ray. Element i = 64 k + lane is handled by `lane` in pass k (coalesced rows); a
// pass does a 6-step wave scan in double and carries the running total of the previous passes in. Double partial sums
// of fp32 inputs are exact unless an addend is < 2^-29 of the running sum, and even then a different association moves
// the double result by ~1e-16 relative, so the fp32-rounded outputs equal the reference's left-to-right double
// accumulation except on a double-rounding tie (probability ~1e-9 per element; the parity tests pin indices
// bit-exactly on their seeds). The previous one-lane-per-ray loop was 5x slower (profiles/).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWaves = 4;  // rays per 256-thread workgroup

// (wave.h: DPP scans - the shuffle-based Hillis-Steele version spent ~1.5 k clocks per scan in ds_bpermute round trips)
__device__ __forceinline__ double wave_scan_inclusive(double v, int /*lane*/) { return wave_scan_inclusive_f64(v); }

// ---------------------------------------------------------------------------------------------------------------
// RaySamples.get_weights (cameras/rays.py:129-152)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void weights_fwd_kernel(const float* __restrict__ t_bins,
                                                               const float* __restrict__ density,
                                                               int64_t num_rays, int S,
                                                               float* __restrict__ weights) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (ray >= num_rays) return;  // wave-uniform
  const float* tb = t_bins + ray * (S + 1);
  const float* dn = density + ray * S;
  float* out = weights + ray * S;
  double carry = 0.0;  // torch.cumsum on CPU: double accumulator, fp32 outputs
  for (int i0 = 0; i0 < S; i0 += 64) {
    const int i = i0 + lane;
    const float dd = i < S ? (tb[i + 1] - tb[i]) * dn[i] : 0.0f;
    const double incl = carry + wave_scan_inclusive((double)dd, lane);
    double excl = wave_shift_up1_f64(incl);
    if (lane == 0) excl = carry;
    carry = wave_read_f64<63>(incl);
    if (i < S) {
      const float alpha = 1.0f - expf(-dd);
      const float trans = expf(-(float)excl);
      out[i] = nan_to_num(alpha * trans);
    }
  }
}

__global__ __launch_bounds__(kThreads) void weights_bwd_kernel(const float* __restrict__ t_bins,
                                                               const float* __restrict__ density,
                                                               const float* __restrict__ dweights,
                                                               int64_t num_rays, int S,
                                                               float* __restrict__ ddensity,
                                                               uint32_t* __restrict__ gate_out,
                                                               uint8_t* __restrict__ ray_mask) {
  extern __shared__ float lds[];  // (ray_bodies.h: weights_bwd_body — per wave ex[S], trans[S], gw[S])
  weights_bwd_body(lds + (size_t)(threadIdx.x >> 6) * 3 * S, t_bins, density, dweights, num_rays, S, ddensity, gate_out, ray_mask);
}

// ---------------------------------------------------------------------------------------------------------------
// PDFSampler.generate_ray_samples (ray_samplers.py:276-372)
// ---------------------------------------------------------------------------------------------------------------
// LDS: per wave  w[S_prev], cdf[S_prev + 1] (+ the S + 1 new edges when they are merged with the existing ones).
// kFused: the launch also does the level's RaySamples.get_weights (weights_out) and, optionally, its median depth
// (DepthRenderer "median", renderers.py:354-364; models/nerfacto.py:346-347 renders one per proposal level) — the three
// launches per proposal level of the training step in one, the weight row never leaves the wave.
template <bool kFused>
__global__ __launch_bounds__(kThreads) void pdf_resample_kernel(
    const float* __restrict__ s_bins_prev, const float* __restrict__ weights, int S_prev,
    const float* __restrict__ u_base, const float* __restrict__ jitter, const float* __restrict__ nears,
    const float* __restrict__ fars, float anneal_host, const float* __restrict__ anneal_dev, float hist_pad, float eps,
    float u_offset, int spacing, int64_t num_rays, int S,
    float* __restrict__ s_bins, float* __restrict__ t_bins, int32_t* __restrict__ inds,
    const float* __restrict__ t_bins_prev, const float* __restrict__ density, float* __restrict__ weights_out,
    float* __restrict__ depth_median, int jitter_per_edge, int include_original) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kWaves + wave;
  if (ray >= num_rays) return;  // wave-uniform; no workgroup barrier below
  const int row_floats = 3 * S_prev + 2 + (include_original ? S + 1 : 0);
  float* w = lds + (size_t)wave * row_floats;
  float* cdf = w + S_prev;
  float* bprev = cdf + S_prev + 1;    // the previous level's spacing-domain edges (gathered by the search below)
  float* fresh = bprev + S_prev + 1;  // include_original only
  PROBE_STAMP(0, 0);
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
  PROBE_STAMP(0, 1);

  if (kFused) {
    // (0) weights of the previous level  (cameras/rays.py:129-152), kept in the LDS row
    const float* tb = t_bins_prev + ray * (S_prev + 1);
    const float* dn = density + ray * S_prev;
    double carry0 = 0.0;
    for (int i0 = 0; i0 < S_prev; i0 += 64) {
      const int i = i0 + lane;
      const float dd = i0 == 0 ? dd_pre[0] : i0 == 64 ? dd_pre[1] : i0 == 128 ? dd_pre[2] : i0 == 192 ? dd_pre[3]
                       : (i < S_prev ? (tb[i + 1] - tb[i]) * dn[i] : 0.0f);  // (selects, not an indexed array: no scratch)
      const double incl = carry0 + wave_scan_inclusive((double)dd, lane);
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
    PROBE_STAMP(0, 2);
    if (depth_median != nullptr) {  // searchsorted(cumsum(w), 0.5, side="left"), clamped
      double carry1 = 0.0;
      int idx = S_prev;
      for (int i0 = 0; i0 < S_prev && idx == S_prev; i0 += 64) {
        const int i = i0 + lane;
        const double incl = carry1 + wave_scan_inclusive(i < S_prev ? (double)w[i] : 0.0, lane);
        carry1 = wave_read_f64<63>(incl);
        const unsigned long long hit = __ballot(i < S_prev && (float)incl >= 0.5f);
        if (hit != 0ull) idx = i0 + __builtin_ctzll(hit);
      }
      idx = min(idx, S_prev - 1);
      if (lane == 0) depth_median[ray] = (tb[idx] + tb[idx + 1]) / 2.0f;
    }
  }
  PROBE_STAMP(0, 3);
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
    total = total + wave_read_f64<63>(wave_scan_inclusive((double)v, lane));
  }
  PROBE_STAMP(0, 4);
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
    const double incl = carry + wave_scan_inclusive((double)pdf, lane);
    carry = wave_read_f64<63>(incl);
    if (i < S_prev) cdf[i + 1] = fminf(1.0f, (float)incl);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  PROBE_STAMP(0, 5);
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
  PROBE_STAMP(0, 6);
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

}  // namespace nsamd

using namespace nsamd;

static inline unsigned ray_blocks(int64_t num_rays) { return (unsigned)((num_rays + kWaves - 1) / kWaves); }

extern "C" int nsamd_piecewise_bins(const float* nears, const float* fars, const float* edges, const float* jitter,
                                    int32_t jitter_per_edge, int64_t num_rays, int32_t S, int spacing, float* s_bins,
                                    float* t_bins, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(nears && fars && edges && s_bins && t_bins);
  const int64_t blocks64 = (num_rays + (kThreads / 64) - 1) / (kThreads / 64);
  if (blocks64 > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  piecewise_bins_kernel<<<(unsigned)blocks64, kThreads, 0, (hipStream_t)stream>>>(
      nears, fars, edges, jitter, jitter_per_edge != 0, num_rays, S, spacing, s_bins, t_bins);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_select_bins(const float* slot_dev, int32_t slots, int64_t num_rays, const float* origins_pool,
                                 const float* directions_pool, const int64_t* cameras_pool, const float* target_pool,
                                 float* origins, float* directions, int64_t* cameras, float* target, const float* nears,
                                 const float* fars, const float* edges, const float* jitter, int32_t jitter_per_edge, int32_t S,
                                 int spacing, float* s_bins, float* t_bins, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && slots >= 1 && S > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(slot_dev && origins_pool && directions_pool && cameras_pool && target_pool && origins && directions &&
                cameras && target && nears && fars && edges && s_bins && t_bins);
  const int64_t blocks64 = (num_rays + (kThreads / 64) - 1) / (kThreads / 64);
  if (blocks64 > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  select_bins_kernel<<<(unsigned)blocks64, kThreads, 0, (hipStream_t)stream>>>(
      slot_dev, slots, num_rays, origins_pool, directions_pool, cameras_pool, target_pool, origins, directions, cameras, target,
      nears, fars, edges, jitter, jitter_per_edge != 0, S, spacing, s_bins, t_bins);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_weights_fwd(const float* t_bins, const float* density, int64_t num_rays, int32_t S,
                                 float* weights, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_bins && density && weights);
  weights_fwd_kernel<<<ray_blocks(num_rays), kThreads, 0, (hipStream_t)stream>>>(t_bins, density, num_rays, S,
                                                                                   weights);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

static int weights_bwd_launch(const float* t_bins, const float* density, const float* dweights, int64_t num_rays,
                              int32_t S, float* ddensity, uint32_t* gate_out, uint8_t* ray_mask, nsamd_stream_t stream,
                              bool gate_precleared = false) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0);
  if (gate_out != nullptr && !gate_precleared &&  // cleared on the stream ahead of the launch (a memset node inside a captured graph)
      hipMemsetAsync(gate_out, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
    return NSAMD_ERR_LAUNCH;
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_bins && density && dweights && ddensity);
  if (S > 1024) return NSAMD_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * 3 * kWaves * (size_t)S;
  weights_bwd_kernel<<<ray_blocks(num_rays), kThreads, lds, (hipStream_t)stream>>>(t_bins, density, dweights,
                                                                                   num_rays, S, ddensity, gate_out, ray_mask);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_weights_bwd(const float* t_bins, const float* density, const float* dweights,
                                 int64_t num_rays, int32_t S, float* ddensity, nsamd_stream_t stream) {
  return weights_bwd_launch(t_bins, density, dweights, num_rays, S, ddensity, nullptr, nullptr, stream);
}

extern "C" int nsamd_weights_bwd_gate(const float* t_bins, const float* density, const float* dweights,
                                      int64_t num_rays, int32_t S, float* ddensity, uint32_t* gate_out,
                                      uint8_t* ray_mask_out, int32_t gate_precleared, nsamd_stream_t stream) {
  NSAMD_REQUIRE(gate_out != nullptr);
  return weights_bwd_launch(t_bins, density, dweights, num_rays, S, ddensity, gate_out, ray_mask_out, stream,
                            gate_precleared != 0);
}

extern "C" int nsamd_pdf_resample(const float* s_bins_prev, const float* weights, int32_t S_prev,
                                  const float* u_base, const float* jitter, const float* nears, const float* fars,
                                  float anneal, const float* anneal_dev, float histogram_padding, float eps,
                                  float u_offset, int spacing, int32_t jitter_per_edge, int32_t include_original,
                                  int64_t num_rays, int32_t S, float* s_bins, float* t_bins, int32_t* inds,
                                  nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0 && S_prev > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(s_bins_prev && weights && u_base && nears && fars && s_bins && t_bins);
  if (S_prev > 1024 || S > 4096) return NSAMD_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * kWaves * (3 * (size_t)S_prev + 2 + (include_original ? (size_t)S + 1 : 0));
  pdf_resample_kernel<false><<<ray_blocks(num_rays), kThreads, lds, (hipStream_t)stream>>>(
      s_bins_prev, weights, S_prev, u_base, jitter, nears, fars, anneal, anneal_dev, histogram_padding, eps, u_offset,
      spacing, num_rays, S, s_bins, t_bins, inds, nullptr, nullptr, nullptr, nullptr, jitter_per_edge != 0,
      include_original != 0);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_proposal_resample(const float* t_bins_prev, const float* s_bins_prev, const float* density,
                                       int32_t S_prev, const float* u_base, const float* jitter, const float* nears,
                                       const float* fars, float anneal, const float* anneal_dev,
                                       float histogram_padding, float eps, float u_offset, int spacing,
                                       int64_t num_rays, int32_t S, float* weights, float* depth_median, float* s_bins,
                                       float* t_bins, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0 && S_prev > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_bins_prev && s_bins_prev && density && u_base && nears && fars && weights && s_bins && t_bins);
  if (S_prev > 1024) return NSAMD_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * kWaves * (3 * (size_t)S_prev + 2);
  pdf_resample_kernel<true><<<ray_blocks(num_rays), kThreads, lds, (hipStream_t)stream>>>(
      s_bins_prev, nullptr, S_prev, u_base, jitter, nears, fars, anneal, anneal_dev, histogram_padding, eps, u_offset,
      spacing, num_rays, S, s_bins, t_bins, nullptr, t_bins_prev, density, weights, depth_median, 0, 0);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
