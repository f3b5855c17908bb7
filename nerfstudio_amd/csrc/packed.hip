// Packed-sample path of instant-ngp for gfx950 (BASELINE configs[3]): occupancy-grid ray
// marching, transmittance scan with visibility-ordered early termination, sample compaction, packed compositing.
// Reference call sites (the arithmetic itself lives in nerfacc 0.5.2, absent from /root/reference — restated, parity of
// the marcher UNPINNED, see oracle/packed_oracle.py):
//   VolumetricSampler.forward              /root/reference/nerfstudio/model_components/ray_samplers.py:385-519
//   NGPModel.get_outputs                   /root/reference/nerfstudio/models/instant_ngp.py:172-217
//   packed branches of the renderers       /root/reference/nerfstudio/model_components/renderers.py:93-102, 310-314, 369-377
//
// Layout: samples of a ray are contiguous, rays in increasing order ("packed"); packed_info[r] = (start, count) int64.
// Variable-length rows, so every per-ray kernel is one WAVEFRONT per ray walking its segment in chunks of 64 with a
// carried running sum (double, like the dense scans of sampler.hip): the scan that gives the transmittance in front of
// each sample is also what decides, in visibility order, where a ray stops contributing.
#include "common.h"

namespace nsamd {

constexpr int kPackThreads = 256;
constexpr int kPackWaves = kPackThreads / 64;

__device__ __forceinline__ double pk_scan_inclusive(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// ---- occupancy-grid marching ---------------------------------------------------------------------------------------
// Multi-level grid as nerfacc's OccGridEstimator lays it out: level l covers the region of interest scaled by 2^l about
// its centre, `resolution`^3 cells each, binaries[level][x][y][z]. A ray marches t = t0, t0 + dt, ... with
// dt = clamp(t * cone_angle, step, 1e10) (cone_angle 0: uniform steps); the sample [t, t + dt) is kept when the cell of
// the FINEST level containing its midpoint is occupied. Sequential fp32 per ray, same op order as the oracle.
__device__ __forceinline__ bool ray_box(const float o[3], const float d[3], const float lo[3], const float hi[3], float& t0,
                                        float& t1) {
  float a = -3.4028234663852886e38f, b = 3.4028234663852886e38f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float inv = 1.0f / d[k];  // +-inf for an axis-parallel ray: IEEE handles the slab test
    float ta = (lo[k] - o[k]) * inv, tb = (hi[k] - o[k]) * inv;
    if (ta > tb) { const float s = ta; ta = tb; tb = s; }
    if (ta != ta || tb != tb) {  // 0 * inf: origin on a slab plane of a parallel axis -> inside iff lo <= o <= hi
      if (o[k] < lo[k] || o[k] > hi[k]) return false;
      continue;
    }
    a = fmaxf(a, ta);
    b = fminf(b, tb);
  }
  t0 = a;
  t1 = b;
  return a <= b;
}

// One WAVEFRONT per ray. The lattice of a ray is defined by the sequential fp32 recurrence
//     t_0 = max(near, box entry) [+ jitter * step],   dt_k = clamp(t_k * cone_angle, step, 1e10),   t_{k+1} = t_k + dt_k
// (what one thread per ray walks in oracle/packed_oracle.py::occgrid_march). 64 consecutive steps are handled per trip:
// lane j re-runs the recurrence j times from the trip's first step (63 masked iterations of 4 dependent VALU ops in
// lock-step: the SAME fp32 operations in the same order as the scalar walk, hence bit-identical t_k), then all 64 lanes
// do their cell lookups at once — ONE memory latency per 64 steps instead of one per step — a ballot gives the kept
// steps, a popcount prefix their packed slots. Round 2 walked a ray per THREAD: 4096 rays = 64 wavefronts on a
// 1024-SIMD chip, each step behind its own L2 round trip.
//
// Empty-space skipping: `coarse` (nullable) is a bitfield with one bit per 4x4x4 block of cells and level
// (nsamd_occgrid_binarise builds it beside the binaries: 4 KiB per level at R = 128), staged in LDS by the workgroup. A
// lane whose block is empty knows its cell is empty without touching the 2 MiB-per-level byte grid — the result is the
// same by construction (an empty block has no occupied cell).
constexpr int kMarchThreads = 512;
constexpr int kMarchWaves = kMarchThreads / 64;
constexpr int kCoarseShift = 2;  // 4^3 cells per coarse block

// `stash` [num_rays, stash_cap] (t_start, t_end) pairs (nullable): the COUNT pass leaves a ray's first stash_cap kept steps
// there, and the WRITE pass copies them into the packed arrays instead of marching the ray a second time (a ray that kept more
// is marched again, as before: same values either way). The march is the dominant packed kernel of the instant-ngp step
// and its two passes were the same 65 us twice.
template <bool kWrite>
__global__ __launch_bounds__(kMarchThreads) void occgrid_march_kernel(
    const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ t_min,
    const float* __restrict__ t_max, int64_t num_rays, float near_plane, float far_plane, nsamd_occgrid grid, float step,
    float cone_angle, const float* __restrict__ jitter, int32_t* __restrict__ counts, const int64_t* __restrict__ starts,
    int64_t* __restrict__ ray_indices, float* __restrict__ t_starts, float* __restrict__ t_ends, int coarse_words,
    float2* __restrict__ stash, int stash_cap) {
  extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
  const int R = grid.resolution, L = grid.levels;
  const bool use_coarse = grid.coarse != nullptr && coarse_words > 0;
  if (kWrite && stash != nullptr) {
    // (before the coarse bits are staged: a workgroup all of whose rays fit their stash never reads the grid)
    const int lane0 = threadIdx.x & 63;
    const int64_t ray0 = (int64_t)blockIdx.x * kMarchWaves + (threadIdx.x >> 6);
    bool remarch = false;
    if (ray0 < num_rays) {
      const int64_t cnt = starts[2 * ray0 + 1], at = starts[2 * ray0];
      if (cnt <= stash_cap) {
        const float2* src = stash + ray0 * stash_cap;
        for (int64_t k = lane0; k < cnt; k += 64) {
          const float2 v = src[k];
          ray_indices[at + k] = ray0;
          t_starts[at + k] = v.x;
          t_ends[at + k] = v.y;
        }
      } else {
        remarch = true;
      }
    }
    if (!__syncthreads_or(remarch)) return;
  }
  if (use_coarse) {
    for (int i = threadIdx.x; i < coarse_words; i += kMarchThreads) coarse_lds[i] = grid.coarse[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kMarchWaves + (threadIdx.x >> 6);
  if (ray >= num_rays) return;  // wave-uniform; no workgroup barrier below
  if (kWrite && stash != nullptr && starts[2 * ray + 1] <= stash_cap) return;  // copied above
  const float o[3] = {origins[3 * ray], origins[3 * ray + 1], origins[3 * ray + 2]};
  const float d[3] = {directions[3 * ray], directions[3 * ray + 1], directions[3 * ray + 2]};
  float t_lo = near_plane, t_hi = far_plane;
  if (t_min != nullptr) t_lo = fmaxf(t_lo, t_min[ray]);
  if (t_max != nullptr) t_hi = fminf(t_hi, t_max[ray]);
  float centre[3], half[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    centre[k] = (grid.aabb[k] + grid.aabb[3 + k]) * 0.5f;
    half[k] = (grid.aabb[3 + k] - grid.aabb[k]) * 0.5f;
  }
  const float outer = (float)(1 << (L - 1));
  float lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = centre[k] - half[k] * outer;
    hi[k] = centre[k] + half[k] * outer;
  }
  const int RC = R >> kCoarseShift;  // coarse blocks per axis
  int32_t n = 0;
  int64_t out = kWrite ? starts[2 * ray] : 0;  // packed_info [N,2]: (start, count)
  float ta, tb;
  if (ray_box(o, d, lo, hi, ta, tb)) {
    float t = fmaxf(t_lo, ta);
    const float t_end = fminf(t_hi, tb);
    if (jitter != nullptr) t += jitter[ray] * step;  // stratified: the whole lattice of the ray shifts by U[0,1) * step
    for (int it = 0; it < (1 << 20) && t < t_end; it += 64) {
      // lane j: t_{it + j} by j applications of the recurrence (all lanes in lock-step, the surplus masked)
      float tj = t;
#pragma unroll 7
      for (int i = 0; i < 63; ++i) {
        float dti = tj * cone_angle;
        dti = fminf(fmaxf(dti, step), 1e10f);
        const float nt = tj + dti;
        tj = i < lane ? nt : tj;
      }
      float dt = tj * cone_angle;
      dt = fminf(fmaxf(dt, step), 1e10f);
      const bool live = tj < t_end && it + lane < (1 << 20);
      bool keep = false;
      if (live) {
        const float mid = tj + dt * 0.5f;
        const float p[3] = {o[0] + d[0] * mid, o[1] + d[1] * mid, o[2] + d[2] * mid};
        // finest level whose box holds the midpoint
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) m = fmaxf(m, fabsf(p[k] - centre[k]) / half[k]);
        int level = 0;
        float scale = 1.0f;
        while (level < L - 1 && m > scale) {
          scale *= 2.0f;
          ++level;
        }
        if (m <= scale) {
          int c[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float u = (p[k] - (centre[k] - half[k] * scale)) / (2.0f * half[k] * scale) * (float)R;
            int ci = (int)floorf(u);
            c[k] = ci < 0 ? 0 : (ci >= R ? R - 1 : ci);
          }
          bool maybe = true;
          if (use_coarse) {
            const int cb = ((level * RC + (c[0] >> kCoarseShift)) * RC + (c[1] >> kCoarseShift)) * RC + (c[2] >> kCoarseShift);
            maybe = (coarse_lds[cb >> 5] >> (cb & 31)) & 1u;
          }
          if (maybe) {
            const size_t cell = (((size_t)level * R + c[0]) * R + c[1]) * R + c[2];
            keep = grid.binaries[cell] != 0;
          }
        }
      }
      const unsigned long long kept = __ballot(keep);
      if (kWrite && keep) {
        const int64_t slot = out + __builtin_popcountll(kept & ((1ull << lane) - 1ull));
        ray_indices[slot] = ray;
        t_starts[slot] = tj;
        t_ends[slot] = tj + dt;
      }
      if (!kWrite && stash != nullptr && keep) {
        const int64_t slot = out + __builtin_popcountll(kept & ((1ull << lane) - 1ull));  // (out = kept so far in count mode)
        if (slot < stash_cap) stash[ray * stash_cap + slot] = make_float2(tj, tj + dt);
      }
      const int c64 = __builtin_popcountll(kept);
      out += c64;
      n += c64;
      // first step of the next trip: t_{it + 64} = t_{it + 63} + dt_{it + 63}
      t = __shfl(tj + dt, 63);
    }
  }
  if (!kWrite && lane == 0) counts[ray] = n;
}

// counts [N] int32 -> packed_info [N,2] int64 (start, count) and the total in total_out[0]; one workgroup.
__global__ __launch_bounds__(1024) void packed_info_kernel(const int32_t* __restrict__ counts, int64_t num_rays,
                                                           int64_t* __restrict__ info, int64_t* __restrict__ total_out) {
  __shared__ long long wave_tot[16];
  __shared__ long long base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < num_rays; i0 += 1024) {
    const int64_t i = i0 + threadIdx.x;
    const long long v = i < num_rays ? counts[i] : 0;
    long long inc = v;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const long long o = __shfl_up(inc, dd);
      if (lane >= dd) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    long long before = base_s;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (i < num_rays) {
      info[2 * i] = before + inc - v;
      info[2 * i + 1] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      long long t = base_s;
      for (int w = 0; w < 16; ++w) t += wave_tot[w];
      base_s = t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) total_out[0] = base_s;
}

// ---- render_weight_from_density / render_visibility_from_density over packed segments ----------------------------
// alpha_i = 1 - exp(-sigma_i dt_i), T_i = exp(-sum_{j<i} sigma_j dt_j), w_i = T_i alpha_i (the formulas of the dense
// RaySamples.get_weights, cameras/rays.py:129-152, without its nan_to_num). kVisibility: instead of the weights, the
// keep-mask T_i >= early_stop_eps && alpha_i >= alpha_thre and its per-ray count — T is non-increasing along the ray, so
// the first sample behind the threshold ends the ray (visibility-ordered early termination).
template <bool kVisibility>
__global__ __launch_bounds__(kPackThreads) void packed_weights_kernel(
    const float* __restrict__ t_starts, const float* __restrict__ t_ends, const float* __restrict__ sigmas,
    const int64_t* __restrict__ info, int64_t num_rays, float early_stop_eps, float alpha_thre,
    float* __restrict__ weights, float* __restrict__ trans_out, uint8_t* __restrict__ mask, int32_t* __restrict__ kept) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
  if (ray >= num_rays) return;  // wave-uniform
  const int64_t s0 = info[2 * ray], cnt = info[2 * ray + 1];
  double carry = 0.0;
  int32_t keep_n = 0;
  for (int64_t i0 = 0; i0 < cnt; i0 += 64) {
    const int64_t i = i0 + lane;
    const bool in = i < cnt;
    const float dd = in ? sigmas[s0 + i] * (t_ends[s0 + i] - t_starts[s0 + i]) : 0.0f;
    const double incl = carry + pk_scan_inclusive((double)dd, lane);
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = carry;
    carry = __shfl(incl, 63);
    const float alpha = 1.0f - expf(-dd);
    const float trans = expf(-(float)excl);
    if (kVisibility) {
      const bool k = in && trans >= early_stop_eps && alpha >= alpha_thre;
      if (in) mask[s0 + i] = k ? 1 : 0;
      keep_n += (int32_t)__builtin_popcountll(__ballot(k));
      // every later sample has a smaller transmittance: nothing behind this chunk can be visible
      if (__shfl(trans, 63) < early_stop_eps && i0 + 64 < cnt) {
        for (int64_t j = i0 + 64 + lane; j < cnt; j += 64) mask[s0 + j] = 0;
        break;
      }
    } else if (in) {
      weights[s0 + i] = trans * alpha;
      if (trans_out != nullptr) trans_out[s0 + i] = trans;
    }
  }
  if (kVisibility && lane == 0) kept[ray] = keep_n;
}

// d w / d sigma: d(sigma_j dt_j) gets  g_j T_j exp(-dd_j)  -  sum_{i>j} g_i w_i  (as the dense weights_bwd_kernel), with
// no per-ray LDS row: pass A sums the segment, pass B walks it backwards — the exclusive prefix is total minus the
// inclusive suffix (double), the suffix of g w comes from the same reverse scan.
__global__ __launch_bounds__(kPackThreads) void packed_weights_bwd_kernel(
    const float* __restrict__ t_starts, const float* __restrict__ t_ends, const float* __restrict__ sigmas,
    const float* __restrict__ dweights, const int64_t* __restrict__ info, int64_t num_rays, float* __restrict__ dsigmas) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  const int64_t s0 = info[2 * ray], cnt = info[2 * ray + 1];
  double total = 0.0;
  for (int64_t i0 = 0; i0 < cnt; i0 += 64) {
    const int64_t i = i0 + lane;
    const float dd = i < cnt ? sigmas[s0 + i] * (t_ends[s0 + i] - t_starts[s0 + i]) : 0.0f;
    total += __shfl(pk_scan_inclusive((double)dd, lane), 63);
  }
  double suf_dd = 0.0, suf_gw = 0.0;  // sums over the samples behind the current chunk
  for (int64_t r0 = 0; r0 < cnt; r0 += 64) {
    const int64_t r = r0 + lane;   // reversed position
    const int64_t i = cnt - 1 - r; // sample
    const bool in = r < cnt;
    const float dt = in ? t_ends[s0 + i] - t_starts[s0 + i] : 0.0f;
    const float dd = in ? sigmas[s0 + i] * dt : 0.0f;
    const double incl_dd = suf_dd + pk_scan_inclusive((double)dd, lane);  // sum over samples >= i
    const float ex = expf(-dd);
    const float trans = expf(-(float)(total - incl_dd));
    const float g = in ? dweights[s0 + i] : 0.0f;
    const float gw = g * ((1.0f - ex) * trans);
    const double incl_gw = suf_gw + pk_scan_inclusive((double)gw, lane);
    double excl_gw = __shfl_up(incl_gw, 1);
    if (lane == 0) excl_gw = suf_gw;
    suf_dd = __shfl(incl_dd, 63);
    suf_gw = __shfl(incl_gw, 63);
    if (in) dsigmas[s0 + i] = dt * (g * trans * ex - (float)excl_gw);
  }
}

// Keep the samples whose mask is set: new packed_info from the per-ray kept counts (packed_info_kernel), then one
// wavefront per ray moves its survivors to their new place (order preserved).
__global__ __launch_bounds__(kPackThreads) void packed_compact_kernel(
    const uint8_t* __restrict__ mask, const int64_t* __restrict__ info_old, const int64_t* __restrict__ info_new,
    int64_t num_rays, const float* __restrict__ t_starts, const float* __restrict__ t_ends,
    int64_t* __restrict__ ray_indices_out, float* __restrict__ t_starts_out, float* __restrict__ t_ends_out) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  const int64_t s0 = info_old[2 * ray], cnt = info_old[2 * ray + 1];
  int64_t dst = info_new[2 * ray];
  for (int64_t i0 = 0; i0 < cnt; i0 += 64) {
    const int64_t i = i0 + lane;
    const bool k = i < cnt && mask[s0 + i] != 0;
    const unsigned long long b = __ballot(k);
    if (k) {
      const int64_t at = dst + __builtin_popcountll(b & ((1ull << lane) - 1ull));
      ray_indices_out[at] = ray;
      t_starts_out[at] = t_starts[s0 + i];
      t_ends_out[at] = t_ends[s0 + i];
    }
    dst += __builtin_popcountll(b);
  }
}

// ---- packed compositing (accumulate_along_rays; renderers.py:93-119, 310-317, 365-383) ------------------------------
// rgb = sum w c (+ background (1 - acc) for "white" / "black" / a colour; "random": nothing), acc = sum w,
// depth = sum w (t_start + t_end) / 2 / (acc + 1e-10). eval_mode: nan_to_num on the samples' colours, clamp to [0, 1].
__global__ __launch_bounds__(kPackThreads) void packed_composite_fwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ weights, const float* __restrict__ t_starts,
    const float* __restrict__ t_ends, const int64_t* __restrict__ info, int64_t num_rays, int bg_mode, float bg0,
    float bg1, float bg2, int eval_mode, float* __restrict__ out_rgb, float* __restrict__ out_acc,
    float* __restrict__ out_depth) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  const int64_t s0 = info[2 * ray], cnt = info[2 * ray + 1];
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, a = 0.f, dsum = 0.f;
  for (int64_t i = lane; i < cnt; i += 64) {
    const float w = weights[s0 + i];
    float r0 = rgb[3 * (s0 + i)], r1 = rgb[3 * (s0 + i) + 1], r2 = rgb[3 * (s0 + i) + 2];
    if (eval_mode) { r0 = nan_to_num(r0); r1 = nan_to_num(r1); r2 = nan_to_num(r2); }
    c0 += w * r0; c1 += w * r1; c2 += w * r2;
    a += w;
    if (t_starts != nullptr) dsum += w * ((t_starts[s0 + i] + t_ends[s0 + i]) / 2.0f);
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    c0 += __shfl_xor(c0, m); c1 += __shfl_xor(c1, m); c2 += __shfl_xor(c2, m);
    a += __shfl_xor(a, m); dsum += __shfl_xor(dsum, m);
  }
  if (lane == 0) {
    if (bg_mode == 1) {  // a constant colour
      c0 = c0 + bg0 * (1.0f - a); c1 = c1 + bg1 * (1.0f - a); c2 = c2 + bg2 * (1.0f - a);
    }
    if (eval_mode) {
      c0 = fminf(fmaxf(c0, 0.0f), 1.0f); c1 = fminf(fmaxf(c1, 0.0f), 1.0f); c2 = fminf(fmaxf(c2, 0.0f), 1.0f);
    }
    out_rgb[3 * ray] = c0; out_rgb[3 * ray + 1] = c1; out_rgb[3 * ray + 2] = c2;
    out_acc[ray] = a;
    if (out_depth != nullptr) out_depth[ray] = dsum / (a + 1e-10f);
  }
}

// gradients of (rgb, acc) w.r.t. the samples: d rgb_s = w_s g_rgb[ray],
// d w_s = g_rgb . c_s + (g_acc - [constant background] g_rgb . bg); depth carries no gradient here (the reference renders
// it under no_grad for the loss path: NGPModel's loss uses rgb only, models/instant_ngp.py:219-235).
__global__ void packed_composite_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ weights,
                                            const int64_t* __restrict__ ray_indices, int64_t n, int bg_mode, float bg0,
                                            float bg1, float bg2, const float* __restrict__ g_rgb,
                                            const float* __restrict__ g_acc, float* __restrict__ d_rgb,
                                            float* __restrict__ d_weights) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int64_t ray = ray_indices[s];
  const float g0 = g_rgb[3 * ray], g1 = g_rgb[3 * ray + 1], g2 = g_rgb[3 * ray + 2];
  const float w = weights[s];
  if (d_rgb != nullptr) {
    d_rgb[3 * s] = w * g0; d_rgb[3 * s + 1] = w * g1; d_rgb[3 * s + 2] = w * g2;
  }
  float dw = g0 * rgb[3 * s] + g1 * rgb[3 * s + 1] + g2 * rgb[3 * s + 2];
  if (g_acc != nullptr) dw += g_acc[ray];
  if (bg_mode == 1) dw -= g0 * bg0 + g1 * bg1 + g2 * bg2;
  d_weights[s] = dw;
}

// positions of packed samples: o[ray] + d[ray] * (t_start + t_end) / 2  (the sigma_fn of VolumetricSampler,
// ray_samplers.py:420-429, and Frustums.get_positions for the packed RaySamples)
__global__ void packed_positions_kernel(const float* __restrict__ origins, const float* __restrict__ directions,
                                        const int64_t* __restrict__ ray_indices, const float* __restrict__ t_starts,
                                        const float* __restrict__ t_ends, int64_t n, float* __restrict__ positions) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int64_t ray = ray_indices[s];
  const float span = t_starts[s] + t_ends[s];
#pragma unroll
  for (int k = 0; k < 3; ++k) positions[3 * s + k] = origins[3 * ray + k] + directions[3 * ray + k] * span / 2.0f;
}

static unsigned pack_ray_blocks(int64_t rays) { return (unsigned)((rays + kPackWaves - 1) / kPackWaves); }

// ---- occupancy-grid maintenance (nerfacc OccGridEstimator.update_every_n_steps / _update) ---------------------------
// x[i] = position inside cell `cells[i]` (flat index over [levels, R, R, R]; nullptr: cell i) at the fractional offset
// jitter[i] in [0,1)^3: level l covers the region of interest scaled by 2^l about its centre.
__global__ __launch_bounds__(256) void occgrid_cell_positions_kernel(const int64_t* __restrict__ cells, int64_t M,
                                                                     nsamd_occgrid grid, const float* __restrict__ jitter,
                                                                     float* __restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int64_t R = grid.resolution, per = R * R * R;
  const int64_t flat = cells != nullptr ? cells[i] : i;
  const int64_t level = flat / per, cell = flat - level * per;
  const int64_t ix[3] = {cell / (R * R), (cell / R) % R, cell % R};
  const float grow = (float)(1 << (int)level);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float u = ((float)ix[k] + jitter[3 * i + k]) / (float)R;  // in [0,1) of the level's box
    const float centre = (grid.aabb[k] + grid.aabb[3 + k]) / 2.0f;
    const float half = (grid.aabb[3 + k] - grid.aabb[k]) / 2.0f * grow;
    x[3 * i + k] = (centre - half) + (u * 2.0f) * half;
  }
}

// occs[c] = max(old[c] * decay, every new estimate of c): `old` is a snapshot of occs taken before the call, so repeated
// cells (the refresh draws random cells WITH replacement) all write the same decayed value, and the estimates — all
// >= 0, so their float bits order like integers — are folded in with an integer atomicMax: the result does not depend
// on the order of the threads. NaN estimates (a diverged field) compare above every number and stay.
__global__ __launch_bounds__(256) void occgrid_decay_kernel(float* __restrict__ occs, const float* __restrict__ old,
                                                            const int64_t* __restrict__ cells, int64_t M, float decay) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int64_t c = cells != nullptr ? cells[i] : i;
  occs[c] = old[c] * decay;
}
__global__ __launch_bounds__(256) void occgrid_max_kernel(float* __restrict__ occs, const int64_t* __restrict__ cells,
                                                          const float* __restrict__ occ_new, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int64_t c = cells != nullptr ? cells[i] : i;
  const float v = fmaxf(occ_new[i], 0.0f);  // (NaN -> 0 under fmaxf; estimates are density x step >= 0)
  atomicMax(reinterpret_cast<int*>(occs) + c, __float_as_int(occ_new[i] != occ_new[i] ? occ_new[i] : v));
}

// mean(occs) in double with a fixed summation order (kSumBlocks contiguous ranges, a fixed tree inside each, the partials
// added in index order), threshold = min(mean, occ_thre) rounded to fp32 once, binaries = occs > threshold, and the
// coarse bitfield the marcher stages in LDS (one bit per 4x4x4 block of cells).
constexpr int kSumBlocks = 1024;
__global__ __launch_bounds__(256) void occgrid_sum_kernel(const float* __restrict__ occs, int64_t total,
                                                          double* __restrict__ partial) {
  __shared__ double red[256];
  const int64_t per = (total + kSumBlocks - 1) / kSumBlocks;
  const int64_t a = (int64_t)blockIdx.x * per, b = a + per < total ? a + per : total;
  double s = 0.0;
  for (int64_t i = a + threadIdx.x; i < b; i += 256) s += (double)occs[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void occgrid_binarise_kernel(const float* __restrict__ occs, int64_t total, float occ_thre,
                                                               const double* __restrict__ partial,
                                                               uint8_t* __restrict__ binaries, float* __restrict__ thre_out) {
  __shared__ float thre_s;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < kSumBlocks; ++k) s += partial[k];
    const double mean = s / (double)total;
    thre_s = (float)(mean < (double)occ_thre ? mean : (double)occ_thre);
    if (blockIdx.x == 0 && thre_out != nullptr) {
      thre_out[0] = thre_s;
      thre_out[1] = (float)mean;
    }
  }
  __syncthreads();
  const float thre = thre_s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    binaries[i] = occs[i] > thre ? 1 : 0;
}
__global__ __launch_bounds__(256) void occgrid_coarse_kernel(const uint8_t* __restrict__ binaries, int levels, int R,
                                                             uint32_t* __restrict__ coarse, int words) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= words) return;
  const int RC = R >> kCoarseShift;
  const int blocks = levels * RC * RC * RC;
  uint32_t bits = 0u;
  for (int b = 0; b < 32; ++b) {
    const int cb = w * 32 + b;
    if (cb >= blocks) break;
    const int level = cb / (RC * RC * RC), r = cb % (RC * RC * RC);
    const int bx = r / (RC * RC), by = (r / RC) % RC, bz = r % RC;
    bool any = false;
    for (int x = 0; x < (1 << kCoarseShift) && !any; ++x)
      for (int y = 0; y < (1 << kCoarseShift) && !any; ++y) {
        // 4 consecutive cells along z = one aligned 32-bit word of the byte grid
        const size_t cell = (((size_t)level * R + (bx << kCoarseShift) + x) * R + (by << kCoarseShift) + y) * R + (bz << kCoarseShift);
        any = *reinterpret_cast<const uint32_t*>(binaries + cell) != 0u;
      }
    bits |= any ? 1u << b : 0u;
  }
  coarse[w] = bits;
}

// words of the coarse bitfield the marcher can use for this grid (0: none — no coarse pointer, a resolution that is not
// a multiple of 4, or more than 64 KiB of bits)
static int coarse_words(const nsamd_occgrid& g) {
  if (g.coarse == nullptr || (g.resolution & ((1 << kCoarseShift) - 1)) != 0) return 0;
  const int64_t rc = g.resolution >> kCoarseShift;
  const int64_t words = ((int64_t)g.levels * rc * rc * rc + 31) / 32;
  return words * 4 <= 64 * 1024 ? (int)words : 0;
}

static int check_grid_desc(const nsamd_occgrid& g) {
  if (g.binaries == nullptr || g.levels < 1 || g.levels > 8 || g.resolution < 1 || g.resolution > 1024)
    return NSAMD_ERR_INVALID_ARG;
  for (int k = 0; k < 3; ++k)
    if (!(g.aabb[3 + k] > g.aabb[k])) return NSAMD_ERR_INVALID_ARG;
  return NSAMD_OK;
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_occgrid_march_count_stash(const float* origins, const float* directions, const float* t_min,
                                               const float* t_max, int64_t num_rays, float near_plane, float far_plane,
                                               nsamd_occgrid grid, float step_size, float cone_angle, const float* jitter,
                                               int32_t* counts, float* stash, int32_t stash_cap, nsamd_stream_t stream) {
  NSAMD_REQUIRE(stash_cap >= 0 && (stash == nullptr || stash_cap > 0));
  NSAMD_REQUIRE(num_rays >= 0 && step_size > 0.0f && cone_angle >= 0.0f);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(origins && directions && counts);
  const int st = check_grid_desc(grid);
  if (st) return st;
  const int64_t nb = (num_rays + kMarchWaves - 1) / kMarchWaves;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  const int cw = coarse_words(grid);
  occgrid_march_kernel<false><<<(unsigned)nb, kMarchThreads, sizeof(uint32_t) * (size_t)cw, (hipStream_t)stream>>>(
      origins, directions, t_min, t_max, num_rays, near_plane, far_plane, grid, step_size, cone_angle, jitter, counts,
      nullptr, nullptr, nullptr, nullptr, cw, reinterpret_cast<float2*>(stash), stash_cap);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_occgrid_march_count(const float* origins, const float* directions, const float* t_min,
                                         const float* t_max, int64_t num_rays, float near_plane, float far_plane,
                                         nsamd_occgrid grid, float step_size, float cone_angle, const float* jitter,
                                         int32_t* counts, nsamd_stream_t stream) {
  return nsamd_occgrid_march_count_stash(origins, directions, t_min, t_max, num_rays, near_plane, far_plane, grid, step_size,
                                         cone_angle, jitter, counts, nullptr, 0, stream);
}

extern "C" int nsamd_occgrid_march_write_stashed(const float* origins, const float* directions, const float* t_min,
                                                 const float* t_max, int64_t num_rays, float near_plane, float far_plane,
                                                 nsamd_occgrid grid, float step_size, float cone_angle, const float* jitter,
                                                 const int64_t* packed_info, const float* stash, int32_t stash_cap,
                                                 int64_t* ray_indices, float* t_starts, float* t_ends, nsamd_stream_t stream) {
  NSAMD_REQUIRE(stash_cap >= 0 && (stash == nullptr || stash_cap > 0));
  NSAMD_REQUIRE(num_rays >= 0 && step_size > 0.0f && cone_angle >= 0.0f);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(origins && directions && packed_info && ray_indices && t_starts && t_ends);
  const int st = check_grid_desc(grid);
  if (st) return st;
  const int64_t nb = (num_rays + kMarchWaves - 1) / kMarchWaves;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  const int cw = coarse_words(grid);
  occgrid_march_kernel<true><<<(unsigned)nb, kMarchThreads, sizeof(uint32_t) * (size_t)cw, (hipStream_t)stream>>>(
      origins, directions, t_min, t_max, num_rays, near_plane, far_plane, grid, step_size, cone_angle, jitter, nullptr,
      packed_info, ray_indices, t_starts, t_ends, cw, const_cast<float2*>(reinterpret_cast<const float2*>(stash)), stash_cap);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_occgrid_march_write(const float* origins, const float* directions, const float* t_min,
                                         const float* t_max, int64_t num_rays, float near_plane, float far_plane,
                                         nsamd_occgrid grid, float step_size, float cone_angle, const float* jitter,
                                         const int64_t* packed_info, int64_t* ray_indices, float* t_starts, float* t_ends,
                                         nsamd_stream_t stream) {
  return nsamd_occgrid_march_write_stashed(origins, directions, t_min, t_max, num_rays, near_plane, far_plane, grid, step_size,
                                           cone_angle, jitter, packed_info, nullptr, 0, ray_indices, t_starts, t_ends, stream);
}

extern "C" int64_t nsamd_occgrid_coarse_words(int32_t levels, int32_t resolution) {
  nsamd_occgrid g{};
  g.levels = levels;
  g.resolution = resolution;
  g.coarse = reinterpret_cast<const uint32_t*>(&g);  // (only tested against nullptr)
  return coarse_words(g);
}

extern "C" int nsamd_occgrid_cell_positions(const int64_t* cells, int64_t M, nsamd_occgrid grid, const float* jitter,
                                            float* positions, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(jitter && positions && grid.levels >= 1 && grid.levels <= 8 && grid.resolution >= 1);
  const int64_t nb = (M + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  occgrid_cell_positions_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(cells, M, grid, jitter, positions);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_occgrid_update(float* occs, int64_t total_cells, const int64_t* cells, const float* occ_new, int64_t M,
                                    float ema_decay, float* scratch, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && total_cells > 0);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(occs && occ_new && scratch);
  const int64_t nb = (M + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  if (hipMemcpyAsync(scratch, occs, sizeof(float) * (size_t)total_cells, hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
      hipSuccess)
    return NSAMD_ERR_LAUNCH;
  occgrid_decay_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(occs, scratch, cells, M, ema_decay);
  NSAMD_CHECK_LAUNCH();
  occgrid_max_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(occs, cells, occ_new, M);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_occgrid_binarise(const float* occs, int32_t levels, int32_t resolution, float occ_thre,
                                      uint8_t* binaries, uint32_t* coarse, double* scratch, float* threshold_out,
                                      nsamd_stream_t stream) {
  NSAMD_REQUIRE(occs && binaries && scratch && levels >= 1 && levels <= 8 && resolution >= 1 && resolution <= 1024);
  const int64_t total = (int64_t)levels * resolution * resolution * resolution;
  occgrid_sum_kernel<<<kSumBlocks, 256, 0, (hipStream_t)stream>>>(occs, total, scratch);
  NSAMD_CHECK_LAUNCH();
  const int64_t nb = (total + 255) / 256;
  occgrid_binarise_kernel<<<(unsigned)(nb < 4096 ? nb : 4096), 256, 0, (hipStream_t)stream>>>(occs, total, occ_thre, scratch,
                                                                                              binaries, threshold_out);
  NSAMD_CHECK_LAUNCH();
  if (coarse != nullptr) {
    const int words = (int)nsamd_occgrid_coarse_words(levels, resolution);
    NSAMD_REQUIRE(words > 0);
    occgrid_coarse_kernel<<<(words + 255) / 256, 256, 0, (hipStream_t)stream>>>(binaries, levels, resolution, coarse, words);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}

extern "C" int nsamd_packed_info(const int32_t* counts, int64_t num_rays, int64_t* packed_info, int64_t* total,
                                 nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && total != nullptr);
  NSAMD_REQUIRE(num_rays == 0 || (counts && packed_info));
  packed_info_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(counts, num_rays, packed_info, total);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_weights_fwd(const float* t_starts, const float* t_ends, const float* sigmas,
                                        const int64_t* packed_info, int64_t num_rays, float* weights, float* transmittance,
                                        nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_starts && t_ends && sigmas && packed_info && weights);
  packed_weights_kernel<false><<<pack_ray_blocks(num_rays), kPackThreads, 0, (hipStream_t)stream>>>(
      t_starts, t_ends, sigmas, packed_info, num_rays, 0.0f, 0.0f, weights, transmittance, nullptr, nullptr);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_weights_bwd(const float* t_starts, const float* t_ends, const float* sigmas,
                                        const float* dweights, const int64_t* packed_info, int64_t num_rays,
                                        float* dsigmas, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_starts && t_ends && sigmas && dweights && packed_info && dsigmas);
  packed_weights_bwd_kernel<<<pack_ray_blocks(num_rays), kPackThreads, 0, (hipStream_t)stream>>>(
      t_starts, t_ends, sigmas, dweights, packed_info, num_rays, dsigmas);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_visibility(const float* t_starts, const float* t_ends, const float* sigmas,
                                       const int64_t* packed_info, int64_t num_rays, float early_stop_eps,
                                       float alpha_thre, uint8_t* mask, int32_t* kept_counts, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(t_starts && t_ends && sigmas && packed_info && mask && kept_counts);
  packed_weights_kernel<true><<<pack_ray_blocks(num_rays), kPackThreads, 0, (hipStream_t)stream>>>(
      t_starts, t_ends, sigmas, packed_info, num_rays, early_stop_eps, alpha_thre, nullptr, nullptr, mask, kept_counts);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_compact(const uint8_t* mask, const int64_t* packed_info_old, const int64_t* packed_info_new,
                                    int64_t num_rays, const float* t_starts, const float* t_ends, int64_t* ray_indices_out,
                                    float* t_starts_out, float* t_ends_out, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(mask && packed_info_old && packed_info_new && t_starts && t_ends && ray_indices_out && t_starts_out &&
                t_ends_out);
  packed_compact_kernel<<<pack_ray_blocks(num_rays), kPackThreads, 0, (hipStream_t)stream>>>(
      mask, packed_info_old, packed_info_new, num_rays, t_starts, t_ends, ray_indices_out, t_starts_out, t_ends_out);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_composite_fwd(const float* rgb, const float* weights, const float* t_starts,
                                          const float* t_ends, const int64_t* packed_info, int64_t num_rays,
                                          int background_mode, const float* background_rgb_host, int eval_mode,
                                          float* out_rgb, float* out_accumulation, float* out_depth,
                                          nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && (background_mode == 0 || background_mode == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rgb && weights && packed_info && out_rgb && out_accumulation);
  NSAMD_REQUIRE(background_mode == 0 || background_rgb_host != nullptr);
  NSAMD_REQUIRE(out_depth == nullptr || (t_starts && t_ends));
  const float b0 = background_mode ? background_rgb_host[0] : 0.f, b1 = background_mode ? background_rgb_host[1] : 0.f,
              b2 = background_mode ? background_rgb_host[2] : 0.f;
  packed_composite_fwd_kernel<<<pack_ray_blocks(num_rays), kPackThreads, 0, (hipStream_t)stream>>>(
      rgb, weights, out_depth ? t_starts : nullptr, t_ends, packed_info, num_rays, background_mode, b0, b1, b2, eval_mode,
      out_rgb, out_accumulation, out_depth);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_composite_bwd(const float* rgb, const float* weights, const int64_t* ray_indices,
                                          int64_t num_samples, int background_mode, const float* background_rgb_host,
                                          const float* g_rgb, const float* g_accumulation, float* d_rgb, float* d_weights,
                                          nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_samples >= 0 && (background_mode == 0 || background_mode == 1));
  if (num_samples == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rgb && weights && ray_indices && g_rgb && d_weights);
  NSAMD_REQUIRE(background_mode == 0 || background_rgb_host != nullptr);
  const float b0 = background_mode ? background_rgb_host[0] : 0.f, b1 = background_mode ? background_rgb_host[1] : 0.f,
              b2 = background_mode ? background_rgb_host[2] : 0.f;
  const int64_t nb = (num_samples + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  packed_composite_bwd_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(rgb, weights, ray_indices, num_samples,
                                                                            background_mode, b0, b1, b2, g_rgb,
                                                                            g_accumulation, d_rgb, d_weights);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_packed_positions(const float* origins, const float* directions, const int64_t* ray_indices,
                                      const float* t_starts, const float* t_ends, int64_t num_samples, float* positions,
                                      nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_samples >= 0);
  if (num_samples == 0) return NSAMD_OK;
  NSAMD_REQUIRE(origins && directions && ray_indices && t_starts && t_ends && positions);
  const int64_t nb = (num_samples + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  packed_positions_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(origins, directions, ray_indices, t_starts,
                                                                        t_ends, num_samples, positions);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
