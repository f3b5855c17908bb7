// The iteration's loss VALUES and training metrics in one small launch (gfx950): the sums over rays of the per-ray terms the
// compositing and loss launches leave behind, in a fixed order (thread t takes rays t, t + 256, ... in double, then a fixed tree),
// scaled as models/nerfacto.py:363-375 does, plus the two training metrics derived from them (models/nerfacto.py:352-361: psnr of
// the rendered colour, distortion). A trainer that logs the losses every step (engine/trainer.py:487-531) reads five floats
// instead of launching a dozen reductions. (Round 5's merged per-ray launch, whose finishing pass this was, is
// csrc/experiments/rounds2to5_opt_in_variants.patch: bit-identical to the launches it replaced and 19 us per iteration slower.)
#include "common.h"

namespace nsamd {

constexpr int kMaxFusedLevels = 4;

struct LossSumArgs {
  const float* sq_err;
  const float* dist_per_ray;
  const float* inter_per_ray[kMaxFusedLevels];
  int levels;
  float rgb_scale, dist_scale, inter_scale, mean_scale;  // 1 / (3 n), mult / n, mult / (n S), 1 / n
  float* out;  // [32]: rgb_loss, interlevel_loss, distortion_loss, psnr, distortion (metric), sum of the three losses, 2 spare;
               // then scratch of the finishing pass: 8 doubles of partial sums and its ticket word (zero before the first launch)
};

__global__ __launch_bounds__(256) void train_loss_values_kernel(int64_t n, LossSumArgs L) {
  __shared__ double s_sum[256];
  if (L.out == nullptr) return;
  // One workgroup per per-ray array (squared error, distortion, interlevel per level), every load of a thread in flight at once,
  // a wave butterfly in double and the four wave sums in wave order: ~3 us for the launch. (The first version summed the arrays
  // one after the other in ONE workgroup with an LDS tree each: ~20 us on the critical path of every iteration that asks for
  // the values — profiles/r05_s9_seam_trace_gaps.txt.) The last workgroup to arrive (a ticket in the scratch words behind the
  // eight result floats) combines the sums in ARRAY order, so the values do not depend on the arrival order.
  const int q = (int)blockIdx.x;
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
  train_loss_values_kernel<<<(unsigned)(2 + levels), 256, 0, (hipStream_t)stream>>>(num_rays, L);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
