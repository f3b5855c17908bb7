// One sample point of a proposal network (HashMLPDensityField.get_density, fields/density_fields.py:94-117): position ->
// contraction / box normalisation -> hash grid (all levels, every gather in flight before the first blend) -> MLP -> trunc_exp.
// The per-point body of density_field_fwd_kernel (density_mlp.hip), shared with the per-ray sampler launch (fused_sampler.hip)
// so that both produce the same bits. Not part of the C ABI.
#pragma once

#include "common.h"
#include "wave.h"

namespace nsamd {

// (x, y, z): the sample's world position; p: its index in the level's [M] arrays. enc_out / selector_out / pre_out nullable.
template <int LEVELS, int H>
__device__ __forceinline__ void density_point(float x, float y, float z, int64_t p, int64_t M, int transform, const nsamd_aabb& box,
                                              const float2* __restrict__ table, const nsamd_grid& grid,
                                              const nsamd_density_mlp& mlp, float* __restrict__ enc_out,
                                              float* __restrict__ selector_out, float* __restrict__ density,
                                              float* __restrict__ pre_out) {
  constexpr int IN = 2 * LEVELS;
  const float sel = normalise_position(transform, box, x, y, z);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  float2 v[LEVELS][8];
  float w[LEVELS][3];
#pragma unroll
  for (int l = 0; l < LEVELS; ++l) {  // all gathers in flight before the first blend
    const Cell c = locate_cell(x, y, z, grid.scalings[l]);
    w[l][0] = c.w[0]; w[l][1] = c.w[1]; w[l][2] = c.w[2];
    const float2* __restrict__ tl = table + ((size_t)l << grid.log2_table_size);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[l][k] = tl[corner_index(c, k, mask)];
  }
  float feat[IN];
#pragma unroll
  for (int l = 0; l < LEVELS; ++l) {
    const float wx = w[l][0], wy = w[l][1], wz = w[l][2];
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      auto g = [&](int k) { return f == 0 ? v[l][k].x : v[l][k].y; };
      const float yc_zc = g(7) * wx + g(6) * ux;
      const float yf_zc = g(5) * wx + g(4) * ux;
      const float yf_zf = g(1) * wx + g(0) * ux;
      const float yc_zf = g(3) * wx + g(2) * ux;
      const float zc = yc_zc * wy + yf_zc * uy;
      const float zf = yc_zf * wy + yf_zf * uy;
      feat[2 * l + f] = zc * wz + zf * uz;
    }
  }
  if (enc_out != nullptr) {
#pragma unroll
    for (int k = 0; k < IN; ++k) enc_out[(int64_t)k * M + p] = feat[k];
  }
  if (selector_out != nullptr) selector_out[p] = sel;
  const float* __restrict__ W0 = mlp.W0;
  const float* __restrict__ b0 = mlp.b0;
  const float* __restrict__ W1 = mlp.W1;
  float out = mlp.b1[0];
#pragma unroll
  for (int j = 0; j < H; ++j) {
    float a = b0[j];
#pragma unroll
    for (int k = 0; k < IN; ++k) a = fmaf(W0[j * IN + k], feat[k], a);
    out = fmaf(W1[j], fmaxf(a, 0.0f), out);
  }
  if (pre_out != nullptr) pre_out[p] = out;
  density[p] = mlp.average_init_density * expf(out) * sel;
}

}  // namespace nsamd
