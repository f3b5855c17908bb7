// The per-ray middle of the nerfacto training iteration in ONE launch (gfx950): everything between the main field's forward
// and its backward is independent across rays — only the loss SUMS cross rays, and those are taken by the host on demand from
// per-ray values. Reference: models/nerfacto.py:298-392 (get_outputs from the field outputs on, get_metrics_dict, get_loss_dict)
// and what autograd runs back through them: cameras/rays.py:129-152 (get_weights), model_components/renderers.py:72-119, 293-317,
// 354-383, model_components/losses.py:31, 53-146.
//
// The training step used to issue, back to back and each waiting for the one before:
//   nsamd_render_train (weights + compositing + MSE)  ->  nsamd_proposal_losses (interlevel x levels, distortion)
//   ->  nsamd_render_train_bwd (compositing backward + weights backward)  [-> nsamd_weights_bwd_gate per proposal level]
// three to five launches of 5-16 us each with ~5 us of dependent-launch gap between them inside the replayed graph. Here the
// grid's y index is a JOB and every job is one wave per ray running the stand-alone launches' device bodies (ray_bodies.h) one
// after the other:
//   job 0            compositing forward (weights, rgb, accumulation, depths, MSE value + gradient)
//                    -> distortion loss (value + gradient on the fine weights) -> compositing backward -> d rgb / d density
//   job 1 + l        interlevel loss of proposal level l against the fine samples (the fine weights are re-derived from the
//                    densities in the wave: same operations, same bits) [-> that level's weights backward: d density of the
//                    proposal samples, the level's gradient flag and per-ray mask]
// What a later body reads of an earlier one's results (the fine weights, the MSE gradient, the distortion gradient, the
// interlevel gradient) goes through global memory exactly as between the launches — written and read by the SAME wave, a fence
// apart — so every output is bit-identical to the separate launches (tests/test_gpu_kernels.py).
// The bodies below read back what the SAME wave stored a moment ago (the MSE gradient of its ray, its weight row): per-ray
// scalars must then come through the vector cache like the stores did — the scalar cache is not coherent with a wave's own
// vector stores inside one launch (a neighbouring ray's earlier read can have left the 64-byte line there). So this translation
// unit keeps the wave index a vector value; the stand-alone launches, which only read what EARLIER launches wrote, use the
// scalar form (wave.h).
#undef NSAMD_SCALAR_RAY
#define NSAMD_SCALAR_RAY 0
#include "ray_bodies.h"

namespace nsamd {

constexpr int kMaxFusedLevels = 4;

struct FusedRayArgs {
  // fine level
  const float* rgb;       // [N,S,3]
  const float* density;   // [N,S]
  const float* t_bins;    // [N,S+1]
  const float* s_bins;    // [N,S+1]
  const float* target;    // [N,3]
  const float* bg_rays;   // [N,3] or null
  float* weights;         // [N,S] out
  float* rgb_out;         // [N,3]
  float* acc;             // [N]
  float* depth_exp;       // [N] (raw; the finishing pass clips)
  float* depth_med;       // [N] or null
  float* ws;              // min / max partials
  float* sq_err;          // [N]
  float* d_rgb_out;       // [N,3]
  float* dist_per_ray;    // [N]
  float* dw_dist;         // [N,S]
  float* d_rgb;           // [N,S,3]
  float* d_density;       // [N,S]
  // proposal levels
  const float* p_s_bins[kMaxFusedLevels];
  const float* p_weights[kMaxFusedLevels];
  float* p_per_ray[kMaxFusedLevels];
  float* p_dw[kMaxFusedLevels];              // null: no gradient for the proposal networks this step
  const float* p_t_bins[kMaxFusedLevels];    // the level's weights backward (null entries: not fused)
  const float* p_density[kMaxFusedLevels];
  float* p_ddensity[kMaxFusedLevels];
  uint32_t* p_gate[kMaxFusedLevels];
  uint8_t* p_mask[kMaxFusedLevels];
  int p_S[kMaxFusedLevels];
  int levels;
  int S;
  int background;
  float bg_r, bg_g, bg_b;
  float mse_scale, inter_scale, dist_scale;
  int row_floats;  // LDS floats per wave: the largest any body of this launch needs
};

__global__ __launch_bounds__(kRenderThreads) void render_losses_train_kernel(FusedRayArgs a, int64_t num_rays) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float blk_min[kRaysPerBlock], blk_max[kRaysPerBlock];
  const int wave = wave_index();
  float* row = lds + (size_t)wave * a.row_floats;
  const int job = blockIdx.y;
  if (job == 0) {
    composite_fwd_body(blk_min, blk_max, a.rgb, nullptr, a.t_bins, num_rays, a.S, a.background, a.bg_r, a.bg_g, a.bg_b, 0,
                       a.rgb_out, a.acc, a.depth_exp, a.depth_med, nullptr, a.ws, a.density, a.weights, a.target, a.mse_scale,
                       a.sq_err, a.d_rgb_out, a.bg_rays);
    // this wave's weight row and MSE gradient are in memory; the next bodies of THIS wave read them back
    __threadfence_block();
    distortion_body(row, a.s_bins, a.weights, a.S, num_rays, a.dist_scale, a.dist_per_ray, a.dw_dist);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();  // (the distortion body's LDS rows are dead; the backward reuses the wave's row)
    composite_bwd_body(row, a.rgb, a.weights, a.t_bins, num_rays, a.S, a.background, a.bg_r, a.bg_g, a.bg_b, a.d_rgb_out, nullptr,
                       nullptr, nullptr, a.dw_dist, a.d_rgb, nullptr, a.density, a.d_density, a.bg_rays);
  } else {
    const int l = job - 1;
    interlevel_body(row, a.s_bins, nullptr, a.S, a.p_s_bins[l], a.p_weights[l], a.p_S[l], num_rays, a.inter_scale, a.p_per_ray[l],
                    a.p_dw[l], a.density, a.t_bins);
    if (a.p_dw[l] != nullptr && a.p_ddensity[l] != nullptr) {
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      weights_bwd_body(row, a.p_t_bins[l], a.p_density[l], a.p_dw[l], num_rays, a.p_S[l], a.p_ddensity[l], a.p_gate[l],
                       a.p_mask[l]);
    }
  }
}

// The launch's finishing pass: the expected depth's global clip (render.hip: depth_clip_kernel, same arithmetic) and — by its
// last workgroup + 1, which has nothing else to do — the iteration's loss VALUES: the sums over rays of the per-ray terms in a
// fixed order (thread t takes rays t, t + 256, ... in double, then a fixed tree), scaled as models/nerfacto.py:363-375 does,
// plus the two training metrics derived from them (models/nerfacto.py:352-361: psnr of the rendered colour, distortion). A
// trainer that logs the losses every step (engine/trainer.py:487-531) reads five floats instead of launching a dozen reductions.
struct LossSumArgs {
  const float* sq_err;
  const float* dist_per_ray;
  const float* inter_per_ray[kMaxFusedLevels];
  int levels;
  float rgb_scale, dist_scale, inter_scale, mean_scale;  // 1 / (3 n), mult / n, mult / (n S), 1 / n
  float* out;  // [32]: rgb_loss, interlevel_loss, distortion_loss, psnr, distortion (metric), sum of the three losses, 2 spare;
               // then scratch of the finishing pass: 8 doubles of partial sums and its ticket word (zero before the first launch)
};

__global__ __launch_bounds__(256) void train_finish_kernel(float* __restrict__ depth, int64_t n, float* __restrict__ ws,
                                                           int partials, int clip_blocks, LossSumArgs L) {
  __shared__ float s_lo[256], s_hi[256];
  __shared__ double s_sum[256];
  if ((int)blockIdx.x < clip_blocks) {
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
    if (blockIdx.x == 0 && threadIdx.x == 0) { ws[0] = lo; ws[1] = hi; }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) depth[i] = fminf(fmaxf(depth[i], lo), hi);
    return;
  }
  if (L.out == nullptr) return;
  // One workgroup per per-ray array (squared error, distortion, interlevel per level), every load of a thread in flight at once,
  // a wave butterfly in double and the four wave sums in wave order: ~3 us for the launch. (The first version summed the arrays
  // one after the other in ONE workgroup with an LDS tree each: ~20 us on the critical path of every iteration that asks for
  // the values — profiles/r05_s9_seam_trace_gaps.txt.) The last workgroup to arrive (a ticket in the scratch words behind the
  // eight result floats) combines the sums in ARRAY order, so the values do not depend on the arrival order.
  const int q = (int)blockIdx.x - clip_blocks;
  const int arrays = 2 + L.levels;
  if (q >= arrays) return;
  const float* src = q == 0 ? L.sq_err : q == 1 ? L.dist_per_ray : L.inter_per_ray[q - 2];
  double acc = 0.0;
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += 256 * 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t i = i0 + (int64_t)u * 256;
      v[u] = src[i < n ? i : n - 1];  // (unconditional loads at a clamped index; dropped below)
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (i0 + (int64_t)u * 256 < n) ? (double)v[u] : 0.0;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double* scratch = reinterpret_cast<double*>(L.out + 8);      // [kMaxFusedLevels + 2] partial sums
  unsigned* ticket = reinterpret_cast<unsigned*>(L.out + 24);  // self-resetting
  scratch[q] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
  __threadfence();
  if (atomicAdd(ticket, 1u) != (unsigned)(arrays - 1)) return;
  __threadfence();
  const volatile double* sc = scratch;
  double inter = 0.0;
  for (int l = 0; l < L.levels; ++l) inter += sc[2 + l];
  const float rgb_loss = (float)sc[0] * L.rgb_scale;
  const float dist_loss = (float)sc[1] * L.dist_scale;
  const float inter_loss = (float)inter * L.inter_scale;
  L.out[0] = rgb_loss;
  L.out[1] = inter_loss;
  L.out[2] = dist_loss;
  L.out[3] = -10.0f * log10f(rgb_loss);
  L.out[4] = (float)sc[1] * L.mean_scale;
  L.out[5] = (rgb_loss + inter_loss) + dist_loss;
  *ticket = 0u;
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_render_losses_train(
    const float* rgb, const float* density, const float* t_bins, const float* s_bins, int64_t num_rays, int32_t S, int background,
    const float* bg_rgb_host, const float* target, float mse_grad_scale, const float* bg_rays, float* weights, float* rgb_out,
    float* acc, float* depth_expected, float* depth_median, float* workspace, float* sq_err, float* d_rgb_out, int32_t levels,
    const float* const* s_bins_prop, const float* const* w_prop, const int32_t* S_prop, float interlevel_grad_scale,
    float distortion_grad_scale, float* const* interlevel_per_ray, float* const* dw_prop, float* distortion_per_ray,
    float* dw_distortion, float* d_rgb, float* d_density, const float* const* t_bins_prop, const float* const* density_prop,
    float* const* ddensity_prop, uint32_t* const* gates, uint8_t* const* ray_masks, float interlevel_loss_mult,
    float distortion_loss_mult, float* loss_values, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && S > 0 && levels >= 0 && levels <= kMaxFusedLevels);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rgb && density && t_bins && s_bins && target && weights && rgb_out && sq_err && d_rgb_out);
  NSAMD_REQUIRE(distortion_per_ray && dw_distortion && d_rgb && d_density);
  NSAMD_REQUIRE(background >= 0 && background <= 3);
  NSAMD_REQUIRE(background != 2 || bg_rgb_host != nullptr);
  NSAMD_REQUIRE(background != 3 || bg_rays != nullptr);
  NSAMD_REQUIRE(depth_expected == nullptr || workspace != nullptr);
  NSAMD_REQUIRE(levels == 0 || (s_bins_prop && w_prop && S_prop && interlevel_per_ray));
  if (S > 1024) return NSAMD_ERR_UNSUPPORTED;
  FusedRayArgs a{};
  a.rgb = rgb, a.density = density, a.t_bins = t_bins, a.s_bins = s_bins, a.target = target, a.bg_rays = bg_rays;
  a.weights = weights, a.rgb_out = rgb_out, a.acc = acc, a.depth_exp = depth_expected, a.depth_med = depth_median;
  a.ws = workspace, a.sq_err = sq_err, a.d_rgb_out = d_rgb_out, a.dist_per_ray = distortion_per_ray, a.dw_dist = dw_distortion;
  a.d_rgb = d_rgb, a.d_density = d_density;
  a.levels = levels, a.S = S, a.background = background;
  a.bg_r = bg_rgb_host ? bg_rgb_host[0] : 0.f, a.bg_g = bg_rgb_host ? bg_rgb_host[1] : 0.f, a.bg_b = bg_rgb_host ? bg_rgb_host[2] : 0.f;
  a.mse_scale = mse_grad_scale, a.inter_scale = interlevel_grad_scale, a.dist_scale = distortion_grad_scale;
  int row = 3 * S;  // compositing backward: dw, ex, trans (the distortion body needs 2 S)
  for (int i = 0; i < levels; ++i) {
    NSAMD_REQUIRE(s_bins_prop[i] && w_prop[i] && interlevel_per_ray[i] && S_prop[i] > 0);
    if (S_prop[i] > 1024) return NSAMD_ERR_UNSUPPORTED;
    a.p_s_bins[i] = s_bins_prop[i], a.p_weights[i] = w_prop[i], a.p_per_ray[i] = interlevel_per_ray[i], a.p_S[i] = S_prop[i];
    a.p_dw[i] = dw_prop ? dw_prop[i] : nullptr;
    const bool wb = a.p_dw[i] != nullptr && ddensity_prop != nullptr && ddensity_prop[i] != nullptr;
    if (wb) {
      NSAMD_REQUIRE(t_bins_prop && density_prop && t_bins_prop[i] && density_prop[i]);
      a.p_t_bins[i] = t_bins_prop[i], a.p_density[i] = density_prop[i], a.p_ddensity[i] = ddensity_prop[i];
      a.p_gate[i] = gates ? gates[i] : nullptr, a.p_mask[i] = ray_masks ? ray_masks[i] : nullptr;
      row = row > 3 * S_prop[i] ? row : 3 * S_prop[i];
    }
    const int ir = interlevel_row_floats(S, S_prop[i]);
    row = row > ir ? row : ir;
  }
  a.row_floats = (row + 3) & ~3;  // rows stay 16-B aligned (the interlevel body keeps doubles at the start of its row)
  const size_t lds = sizeof(float) * (size_t)a.row_floats * kRaysPerBlock;
  if (lds > 64 * 1024) return NSAMD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((num_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  dim3 g(blocks, (unsigned)(levels + 1));
  render_losses_train_kernel<<<g, kRenderThreads, lds, st>>>(a, num_rays);
  NSAMD_CHECK_LAUNCH();
  if (depth_expected != nullptr || loss_values != nullptr) {
    LossSumArgs L{};
    L.sq_err = sq_err, L.dist_per_ray = distortion_per_ray, L.levels = levels, L.out = loss_values;
    for (int i = 0; i < levels; ++i) L.inter_per_ray[i] = interlevel_per_ray[i];
    L.rgb_scale = 1.0f / (3.0f * (float)num_rays);
    L.dist_scale = distortion_loss_mult / (float)num_rays;
    L.inter_scale = interlevel_loss_mult / ((float)num_rays * (float)S);
    L.mean_scale = 1.0f / (float)num_rays;
    const int clip_blocks = depth_expected != nullptr ? (int)((num_rays + 255) / 256) : 0;
    train_finish_kernel<<<(unsigned)(clip_blocks + (loss_values != nullptr ? 2 + levels : 0)), 256, 0, st>>>(
        depth_expected, num_rays, workspace, (int)blocks, clip_blocks, L);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}

// The iteration's loss values and training metrics alone (the finishing pass of nsamd_render_losses_train without a launch in
// front of it): for a training step that runs the compositing, the losses and the compositing backward as separate launches and
// whose caller logs the losses every iteration (pipeline.TrainEngine under the reference's Trainer.train_iteration).
extern "C" int nsamd_train_loss_values(const float* sq_err, const float* distortion_per_ray, int32_t levels,
                                       const float* const* interlevel_per_ray, int64_t num_rays, int32_t S,
                                       float interlevel_loss_mult, float distortion_loss_mult, float* loss_values,
                                       nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays > 0 && S > 0 && levels >= 0 && levels <= kMaxFusedLevels);
  NSAMD_REQUIRE(sq_err && distortion_per_ray && loss_values && (levels == 0 || interlevel_per_ray));
  LossSumArgs L{};
  L.sq_err = sq_err, L.dist_per_ray = distortion_per_ray, L.levels = levels, L.out = loss_values;
  for (int i = 0; i < levels; ++i) {
    NSAMD_REQUIRE(interlevel_per_ray[i] != nullptr);
    L.inter_per_ray[i] = interlevel_per_ray[i];
  }
  L.rgb_scale = 1.0f / (3.0f * (float)num_rays);
  L.dist_scale = distortion_loss_mult / (float)num_rays;
  L.inter_scale = interlevel_loss_mult / ((float)num_rays * (float)S);
  L.mean_scale = 1.0f / (float)num_rays;
  train_finish_kernel<<<(unsigned)(2 + levels), 256, 0, (hipStream_t)stream>>>(nullptr, num_rays, nullptr, 0, 0, L);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
