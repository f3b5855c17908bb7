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

#include "proposal_chain.h"
#include "ray_bodies.h"

NSAMD_PROBE_DEFINE(sampler)

namespace nsamd {

constexpr int kThreads = 256;  // 4 wavefronts

// ---------------------------------------------------------------------------------------------------------------
// UniformLinDispPiecewiseSampler (ray_samplers.py:78-128, 225-248): pure elementwise.
// ---------------------------------------------------------------------------------------------------------------
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
  const int64_t ray = (int64_t)blockIdx.x * (kThreads / 64) + wave_index();
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
  const int64_t ray = (int64_t)blockIdx.x * (kThreads / 64) + wave_index();
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
  const int64_t ray = (int64_t)blockIdx.x * kWaves + wave_index();
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
  weights_bwd_body(lds + (size_t)wave_index() * 3 * S, t_bins, density, dweights, num_rays, S, ddensity, gate_out, ray_mask);
}

// two levels' weights backward in one launch (proposal_chain.h): blockIdx.y selects the call, each with its own sample count
__global__ __launch_bounds__(kThreads) void weights_bwd_pair_kernel(WeightsBwdCall a, WeightsBwdCall b) {
  extern __shared__ float lds[];
  if (blockIdx.y == 0)
    weights_bwd_body(lds + (size_t)wave_index() * 3 * a.S, a.t_bins, a.density, a.dweights, a.num_rays, a.S, a.ddensity, a.gate,
                     a.ray_mask);
  else
    weights_bwd_body(lds + (size_t)wave_index() * 3 * b.S, b.t_bins, b.density, b.dweights, b.num_rays, b.S, b.ddensity, b.gate,
                     b.ray_mask);
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
  extern __shared__ float lds[];  // (ray_bodies.h: pdf_resample_body)
  const int64_t ray = (int64_t)blockIdx.x * kWaves + wave_index();
  if (ray >= num_rays) return;  // wave-uniform; no workgroup barrier in the body
  const int row_floats = 3 * S_prev + 2 + (include_original ? S + 1 : 0);
  pdf_resample_body<kFused>(lds + (size_t)wave_index() * row_floats, ray, s_bins_prev, weights, S_prev, u_base, jitter, nears,
                            fars, anneal_host, anneal_dev, hist_pad, eps, u_offset, spacing, num_rays, S, s_bins, t_bins, inds,
                            t_bins_prev, density, weights_out, depth_median, jitter_per_edge, include_original);
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

namespace nsamd {
int weights_bwd_launch_pair(const WeightsBwdCall& a, const WeightsBwdCall& b, hipStream_t stream) {
  // (the gates are cleared by the caller)
  if (a.num_rays <= 0 || b.num_rays <= 0 || a.S <= 0 || b.S <= 0 || a.S > 1024 || b.S > 1024) return NSAMD_ERR_UNSUPPORTED;
  if (!(a.t_bins && a.density && a.dweights && a.ddensity && b.t_bins && b.density && b.dweights && b.ddensity))
    return NSAMD_ERR_INVALID_ARG;
  const int64_t rays = a.num_rays > b.num_rays ? a.num_rays : b.num_rays;
  const size_t lds = sizeof(float) * 3 * kWaves * (size_t)(a.S > b.S ? a.S : b.S);
  weights_bwd_pair_kernel<<<dim3(ray_blocks(rays), 2u), kThreads, lds, stream>>>(a, b);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
}  // namespace nsamd

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
