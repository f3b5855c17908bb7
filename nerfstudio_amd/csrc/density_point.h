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

// The same point, with the hash gathers of a LANE PAIR (2i, 2i + 1) arranged so that the two x-neighbours of a cell edge are
// fetched by adjacent lanes of ONE instruction (csrc/hashgrid.hip, hash_encode_fwd_v3_kernel: a divergent gather costs the
// number of distinct 128-B lines an instruction touches, and the two entries share a line 15 times out of 16): per level, four
// instructions on the even lane's point (even lane: lo-x corners, odd lane: hi-x corners), four on the odd lane's — the same 8
// gathers per lane and level, half the lines. Cell hashes and fetched values cross the pair as DPP moves; the blend, the MLP
// and every store are the lane's own point's, operation for operation as above: same bits. Every lane of the wave must call it
// (`live` = the lane has a point: its stores are made).
template <int LEVELS, int H>
__device__ __forceinline__ void density_point_paired(float x, float y, float z, int64_t p, int64_t M, bool live, int transform,
                                                     const nsamd_aabb& box, const float2* __restrict__ table, const nsamd_grid& grid,
                                                     const nsamd_density_mlp& mlp, float* __restrict__ enc_out,
                                                     float* __restrict__ selector_out, float* __restrict__ density,
                                                     float* __restrict__ pre_out) {
  constexpr int IN = 2 * LEVELS;
  const float sel = normalise_position(transform, box, x, y, z);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const bool odd = (threadIdx.x & 1u) != 0u;
  float2 e[LEVELS][4], o[LEVELS][4];  // round E (the even lane's point) / round O (the odd lane's): this lane's x side, 4 (y, z) corners
  float w[LEVELS][3];
#pragma unroll
  for (int l = 0; l < LEVELS; ++l) {  // all gathers in flight before the first exchange
    const Cell c = locate_cell(x, y, z, grid.scalings[l]);
    w[l][0] = c.w[0]; w[l][1] = c.w[1]; w[l][2] = c.w[2];
    const float2* __restrict__ tl = table + ((size_t)l << grid.log2_table_size);
    const uint32_t hy0 = (uint32_t)c.lo[1] * kPrimeY, hy1 = (uint32_t)c.hi[1] * kPrimeY;
    const uint32_t hz0 = (uint32_t)c.lo[2] * kPrimeZ, hz1 = (uint32_t)c.hi[2] * kPrimeZ;
    const uint32_t nx = pair_swap_u32(odd ? (uint32_t)c.lo[0] : (uint32_t)c.hi[0]);  // the neighbour's point, THIS lane's x side
    const uint32_t ny0 = pair_swap_u32(hy0), ny1 = pair_swap_u32(hy1), nz0 = pair_swap_u32(hz0), nz1 = pair_swap_u32(hz1);
    const uint32_t own_x = odd ? (uint32_t)c.hi[0] : (uint32_t)c.lo[0];
    const uint32_t ex = odd ? nx : own_x, ey0 = odd ? ny0 : hy0, ey1 = odd ? ny1 : hy1, ez0 = odd ? nz0 : hz0, ez1 = odd ? nz1 : hz1;
    const uint32_t ox = odd ? own_x : nx, oy0 = odd ? hy0 : ny0, oy1 = odd ? hy1 : ny1, oz0 = odd ? hz0 : nz0, oz1 = odd ? hz1 : nz1;
    e[l][0] = tl[(ex ^ ey0 ^ ez0) & mask]; e[l][1] = tl[(ex ^ ey1 ^ ez0) & mask];
    e[l][2] = tl[(ex ^ ey0 ^ ez1) & mask]; e[l][3] = tl[(ex ^ ey1 ^ ez1) & mask];
    o[l][0] = tl[(ox ^ oy0 ^ oz0) & mask]; o[l][1] = tl[(ox ^ oy1 ^ oz0) & mask];
    o[l][2] = tl[(ox ^ oy0 ^ oz1) & mask]; o[l][3] = tl[(ox ^ oy1 ^ oz1) & mask];
  }
  float feat[IN];
#pragma unroll
  for (int l = 0; l < LEVELS; ++l) {
    float2 v[8];  // corner k: bit0 = x is ceil, bit1 = y, bit2 = z (corner_index)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2 send = odd ? e[l][q] : o[l][q];  // what was fetched for the neighbour's point
      const float2 r = make_float2(pair_swap_f32(send.x), pair_swap_f32(send.y));
      v[2 * q] = odd ? r : e[l][q];      // lo-x corner of the own point
      v[2 * q + 1] = odd ? o[l][q] : r;  // hi-x corner
    }
    const float wx = w[l][0], wy = w[l][1], wz = w[l][2];
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      auto g = [&](int k) { return f == 0 ? v[k].x : v[k].y; };
      const float yc_zc = g(7) * wx + g(6) * ux;
      const float yf_zc = g(5) * wx + g(4) * ux;
      const float yf_zf = g(1) * wx + g(0) * ux;
      const float yc_zf = g(3) * wx + g(2) * ux;
      const float zc = yc_zc * wy + yf_zc * uy;
      const float zf = yc_zf * wy + yf_zf * uy;
      feat[2 * l + f] = zc * wz + zf * uz;
    }
  }
  if (!live) return;
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
