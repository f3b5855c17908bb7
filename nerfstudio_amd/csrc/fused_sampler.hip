// ProposalNetworkSampler.generate_ray_samples (model_components/ray_samplers.py:576-617) in ONE launch (gfx950): the whole
// sampling cascade of a ray is independent of every other ray —
//   [batch selection ->] initial bins (UniformLinDispPiecewiseSampler, :78-128, 225-248)
//   -> for each proposal level: density of its samples (HashMLPDensityField.get_density, fields/density_fields.py:94-117)
//      -> weights (cameras/rays.py:129-152) [-> median depth] -> PDF resampling (:276-372) of the next level's bin edges
// — so one wavefront walks one ray through all of it. The training step used to issue 1 + 2 launches per level, each waiting
// for the one before (five dependent launches and their gaps for nerfacto's 256 -> 96 -> 48); the density launches are bound by
// their VALU work (919 vector instructions per 64 points: 28 us for the 1 M points of level 0) and the resampling launches by
// one wave's chain of latencies — inside one launch the second hides behind the first of the waves beside it, and the
// per-ray scalars of a point's position (origin, direction) are wave-uniform instead of eight loads per point.
// The stages are the stand-alone launches' device bodies (ray_bodies.h, density_point.h) run one after the other; what one
// stage hands the next (bin edges, densities) goes through global memory exactly as between the launches, written and read by
// the SAME wave a fence apart: every output is bit-identical to the separate launches (tests/test_gpu_fused_launches.py).
#include "density_point.h"
#include "ray_bodies.h"

namespace nsamd {

constexpr int kMaxSamplerLevels = 4;

struct SamplerLevel {
  const float2* table;
  nsamd_grid grid;
  nsamd_density_mlp mlp;
  nsamd_aabb box;
  int transform;
  int S;              // samples of this level
  float* s_bins;      // [N, S+1] this level's edges (level 0: written by the bins stage; l > 0: by level l-1's resampling)
  float* t_bins;
  float* density;     // [N*S] out
  float* enc;         // [2 LEVELS, N*S] out, nullable (kept for the level's backward on the steps that update it)
  float* selector;    // [N*S] out, nullable
  float* pre;         // [N*S] out, nullable
  float* weights;     // [N, S] out
  float* depth_med;   // [N] out, nullable
  const float* u_base;   // [S_next + 1] of the resampling that follows this level
  const float* jitter;   // [N] draw of that resampling
  float u_offset;        // 1 / (2 (S_next + 1))
};

struct SamplerArgs {
  // batch selection (nullable: the caller filled the ray buffers)
  const float* slot_dev;
  int slots;
  const float* origins_pool;
  const float* directions_pool;
  const int64_t* cameras_pool;
  const float* target_pool;
  int64_t* cameras;
  float* target;
  // rays
  float* origins;
  float* directions;
  const float* nears;
  const float* fars;
  // initial bins
  const float* edges;
  const float* jitter0;
  int spacing;
  // resampling
  const float* anneal_dev;
  float anneal_host, hist_pad, eps;
  // levels
  int levels;
  SamplerLevel L[kMaxSamplerLevels];
  // the final level's edges (written by the last resampling)
  float* s_bins_out;
  float* t_bins_out;
  int S_out;
  int row_floats;  // LDS floats per wave
};

template <int LEVELS, int H>
__global__ __launch_bounds__(kRenderThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void proposal_sampler_kernel(SamplerArgs a, int64_t num_rays) {
  // (four waves per SIMD = every ray of a 4096-ray batch resident at once; the wave index is made a scalar so that everything
  //  derived from the ray — its origin, direction, row pointers — lives in scalar registers)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_index();
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  if (ray >= num_rays) return;  // wave-uniform; no workgroup barrier below
  float* row = lds + (size_t)wave * a.row_floats;
  // ---- [batch selection] + initial bins (sampler.hip: select_bins_kernel) ----
  int32_t sel_slot = 0;
  if (a.slot_dev != nullptr) {
    int32_t slot = (int32_t)a.slot_dev[0];
    slot = slot < 0 ? 0 : (slot >= a.slots ? a.slots - 1 : slot);
    sel_slot = slot;
    if (lane < 3) {
      const int64_t dst = 3 * ray + lane, src = (int64_t)slot * 3 * num_rays + dst;
      a.origins[dst] = a.origins_pool[src];
      a.directions[dst] = a.directions_pool[src];
      a.target[dst] = a.target_pool[src];
    } else if (lane == 3) {
      a.cameras[ray] = a.cameras_pool[(int64_t)slot * num_rays + ray];
    }
  }
  piecewise_bins_body(ray, lane, a.nears, a.fars, a.edges, a.jitter0, 0, a.L[0].S, a.spacing, a.L[0].s_bins, a.L[0].t_bins);
  __threadfence_block();  // this wave reads its ray and its bin edges back below
#pragma unroll 1
  for (int l = 0; l < a.levels; ++l) {
    const SamplerLevel& Lv = a.L[l];
    const int S = Lv.S;
    const int64_t M = num_rays * S;
    // ---- density of the level's samples: one point per lane and pass; origin and direction are the wave's own ray ----
    // (the ray as this launch's INPUT holds it — the pool slot when the launch selects the batch itself: the copy in
    //  `origins` was stored by this wave's vector lanes a moment ago, and a scalar load of it could hit a line that a
    //  neighbouring ray's wave left in the scalar cache before the store)
    const float* o = (a.slot_dev != nullptr ? a.origins_pool + (int64_t)sel_slot * 3 * num_rays : a.origins) + 3 * ray;
    const float* d = (a.slot_dev != nullptr ? a.directions_pool + (int64_t)sel_slot * 3 * num_rays : a.directions) + 3 * ray;
    const float o0 = o[0], o1 = o[1], o2 = o[2], d0 = d[0], d1 = d[1], d2 = d[2];
    const float* tb = Lv.t_bins + ray * (S + 1);
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      if (s < S) {
        // Frustums.get_positions (common.h: load_position): o + d * (start + end) / 2
        const float span = tb[s] + tb[s + 1];
        const float x = o0 + d0 * span / 2.0f;
        const float y = o1 + d1 * span / 2.0f;
        const float z = o2 + d2 * span / 2.0f;
        density_point<LEVELS, H>(x, y, z, ray * S + s, M, Lv.transform, Lv.box, Lv.table, Lv.grid, Lv.mlp, Lv.enc, Lv.selector,
                                 Lv.density, Lv.pre);
      }
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // ---- weights, median depth, and the next level's bin edges (sampler.hip: pdf_resample_kernel<true>) ----
    const bool last = l + 1 == a.levels;
    const int S_next = last ? a.S_out : a.L[l + 1].S;
    pdf_resample_body<true>(row, ray, Lv.s_bins, nullptr, S, Lv.u_base, Lv.jitter, a.nears, a.fars, a.anneal_host, a.anneal_dev,
                            a.hist_pad, a.eps, Lv.u_offset, a.spacing, num_rays, S_next, last ? a.s_bins_out : a.L[l + 1].s_bins,
                            last ? a.t_bins_out : a.L[l + 1].t_bins, nullptr, Lv.t_bins, Lv.density, Lv.weights, Lv.depth_med, 0, 0);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nsamd

using namespace nsamd;

// Host-side mirror of SamplerLevel for the C ABI (include/nsamd.h: nsamd_sampler_level)
extern "C" int nsamd_proposal_sampler(const float* slot_dev, int32_t slots, const float* origins_pool,
                                      const float* directions_pool, const int64_t* cameras_pool, const float* target_pool,
                                      int64_t* cameras, float* target, float* origins, float* directions, const float* nears,
                                      const float* fars, int64_t num_rays, const float* edges, const float* jitter0, int spacing,
                                      float anneal, const float* anneal_dev, float histogram_padding, float eps, int32_t levels,
                                      const nsamd_sampler_level* level, int32_t S_out, float* s_bins_out, float* t_bins_out,
                                      nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && levels >= 1 && levels <= kMaxSamplerLevels && S_out > 0 && (spacing == 0 || spacing == 1));
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(level && origins && directions && nears && fars && edges && s_bins_out && t_bins_out);
  if (slot_dev != nullptr)
    NSAMD_REQUIRE(slots >= 1 && origins_pool && directions_pool && cameras_pool && target_pool && cameras && target);
  SamplerArgs a{};
  a.slot_dev = slot_dev, a.slots = slots, a.origins_pool = origins_pool, a.directions_pool = directions_pool;
  a.cameras_pool = cameras_pool, a.target_pool = target_pool, a.cameras = cameras, a.target = target;
  a.origins = origins, a.directions = directions, a.nears = nears, a.fars = fars, a.edges = edges, a.jitter0 = jitter0;
  a.spacing = spacing, a.anneal_dev = anneal_dev, a.anneal_host = anneal, a.hist_pad = histogram_padding, a.eps = eps;
  a.levels = levels, a.s_bins_out = s_bins_out, a.t_bins_out = t_bins_out, a.S_out = S_out;
  int row = 0;
  const int num_levels = level[0].grid.num_levels, hidden = level[0].mlp.hidden;
  for (int l = 0; l < levels; ++l) {
    const nsamd_sampler_level& src = level[l];
    NSAMD_REQUIRE(src.table && src.mlp.W0 && src.mlp.b0 && src.mlp.W1 && src.mlp.b1 && src.s_bins && src.t_bins && src.density &&
                  src.weights && src.u_base && src.samples > 0 && src.transform >= 0 && src.transform <= 2);
    // one instantiation per launch: every level's network has the same shape (the nerfacto recipes do), a shape the fused
    // density kernel is built for, and rows the resampling stage holds in LDS
    if (src.grid.num_levels != num_levels || src.mlp.hidden != hidden || src.mlp.in_dim != 2 * num_levels) return NSAMD_ERR_UNSUPPORTED;
    if (src.grid.log2_table_size < 1 || src.grid.log2_table_size > 28 || src.samples > 1024) return NSAMD_ERR_UNSUPPORTED;
    SamplerLevel& L = a.L[l];
    L.table = reinterpret_cast<const float2*>(src.table), L.grid = src.grid, L.mlp = src.mlp, L.box = src.aabb;
    L.transform = src.transform, L.S = src.samples, L.s_bins = src.s_bins, L.t_bins = src.t_bins, L.density = src.density;
    L.enc = src.enc, L.selector = src.selector, L.pre = src.pre, L.weights = src.weights, L.depth_med = src.depth_median;
    L.u_base = src.u_base, L.jitter = src.jitter, L.u_offset = src.u_offset;
    const int r = 3 * src.samples + 2;
    row = row > r ? row : r;
  }
  a.row_floats = (row + 3) & ~3;
  const size_t lds = sizeof(float) * (size_t)a.row_floats * kRaysPerBlock;
  const int64_t blocks64 = (num_rays + kRaysPerBlock - 1) / kRaysPerBlock;
  if (blocks64 > 0x7fffffffLL || lds > 64 * 1024) return NSAMD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (num_levels == 5 && hidden == 16) {
    proposal_sampler_kernel<5, 16><<<(unsigned)blocks64, kRenderThreads, lds, st>>>(a, num_rays);
  } else if (num_levels == 5 && hidden == 64) {
    proposal_sampler_kernel<5, 64><<<(unsigned)blocks64, kRenderThreads, lds, st>>>(a, num_rays);
  } else {
    return NSAMD_ERR_UNSUPPORTED;  // callers fall back to the per-level launches
  }
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
