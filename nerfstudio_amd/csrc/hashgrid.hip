// Multiresolution hash encoding for gfx950, torch-path semantics of the reference
// (HashEncoding.pytorch_fwd, /root/reference/nerfstudio/field_components/encodings.py:417-458).
//
// Mapping: grid = (ceil(M/256), L). blockIdx.y is the LEVEL, so the blocks resident at any moment sweep one or two
// levels of the table (2^log2_T * 8 B each; 4 MiB for the nerfacto main grid = one XCD's L2) instead of all L —
// the gathers are L2 hits instead of Infinity-Cache/HBM trips. Consecutive lanes are consecutive samples of one
// ray, i.e. spatial neighbours: on the coarse levels most of a wavefront reads the same few 128-B lines.
// The feature-major output (stride_p = 1) makes the 8-B-per-point result store one coalesced 256-B row per
// wavefront and feature; the [M, 2L] row-major layout of the stand-alone Encoding API is the strided variant.
//
// HBM-bound integer/gather work: no LDS, no MFMA. Algorithmic bytes: 8 corners x 8 B per point and level (fwd),
// 8 corners x 2 atomics x 4 B (bwd).
#include "common.h"

namespace nsamd {

constexpr int kHashBlock = 256;

__global__ __launch_bounds__(kHashBlock) void hash_encode_fwd_kernel(nsamd_points P, int64_t M, int transform,
                                                                     nsamd_aabb box,
                                                                     const float2* __restrict__ table,
                                                                     nsamd_grid grid, float* __restrict__ enc,
                                                                     int64_t stride_p, int64_t stride_k,
                                                                     float* __restrict__ selector) {
  const int level = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * kHashBlock + threadIdx.x;
  if (p >= M) return;
  float x, y, z;
  load_position(P, p, x, y, z);
  const float sel = normalise_position(transform, box, x, y, z);
  if (level == 0 && selector != nullptr) selector[p] = sel;

  const Cell c = locate_cell(x, y, z, grid.scalings[level]);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const float2* __restrict__ tl = table + ((size_t)level << grid.log2_table_size);
  float2 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = tl[corner_index(c, k, mask)];

  const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
  const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
  float r[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    auto g = [&](int k) { return f == 0 ? v[k].x : v[k].y; };
    // blend order x, y, z exactly as encodings.py:446-456
    const float yc_zc = g(7) * wx + g(6) * ux;
    const float yf_zc = g(5) * wx + g(4) * ux;
    const float yf_zf = g(1) * wx + g(0) * ux;
    const float yc_zf = g(3) * wx + g(2) * ux;
    const float zc = yc_zc * wy + yf_zc * uy;
    const float zf = yc_zf * wy + yf_zf * uy;
    r[f] = zc * wz + zf * uz;
  }
  float* o = enc + p * stride_p + (int64_t)(2 * level) * stride_k;
  o[0] = r[0];
  o[stride_k] = r[1];
}

// dL/dtable: one thread per (point, level); 16 fire-and-forget fp32 atomics (global_atomic_add_f32).
__global__ __launch_bounds__(kHashBlock) void hash_encode_bwd_table_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, float* __restrict__ dtable) {
  const int level = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * kHashBlock + threadIdx.x;
  if (p >= M) return;
  const float* gptr = denc + p * stride_p + (int64_t)(2 * level) * stride_k;
  const float g0 = gptr[0], g1 = gptr[stride_k];
  if (g0 == 0.0f && g1 == 0.0f) return;  // adding zero is a no-op; skips masked / zero-weight samples
  float x, y, z;
  load_position(P, p, x, y, z);
  (void)normalise_position(transform, box, x, y, z);
  const Cell c = locate_cell(x, y, z, grid.scalings[level]);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  float* tl = dtable + (((size_t)level << grid.log2_table_size) << 1);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // autograd order: ((g * wz) * wy) * wx
    const float bz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
    const float by = (k & 2) ? c.w[1] : 1.0f - c.w[1];
    const float bx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
    const uint32_t idx = corner_index(c, k, mask);
    unsafeAtomicAdd(tl + 2 * (size_t)idx + 0, ((g0 * bz) * by) * bx);
    unsafeAtomicAdd(tl + 2 * (size_t)idx + 1, ((g1 * bz) * by) * bx);
  }
}

// dL/dposition: one thread per point, loops the levels (no atomics). Only needed when the camera optimiser or
// normals are on (SURVEY.md §8a gradient-flow facts).
__global__ __launch_bounds__(kHashBlock) void hash_encode_bwd_pos_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, const float2* __restrict__ table, nsamd_grid grid,
    const float* __restrict__ denc, int64_t stride_p, int64_t stride_k, float* __restrict__ dpos) {
  const int64_t p = (int64_t)blockIdx.x * kHashBlock + threadIdx.x;
  if (p >= M) return;
  float rx, ry, rz;
  load_position(P, p, rx, ry, rz);
  float x = rx, y = ry, z = rz;
  const float sel = normalise_position(transform, box, x, y, z);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  for (int level = 0; level < grid.num_levels; ++level) {
    const float scale = grid.scalings[level];
    const Cell c = locate_cell(x, y, z, scale);
    const float2* __restrict__ tl = table + ((size_t)level << grid.log2_table_size);
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tl[corner_index(c, k, mask)];
    const float* gptr = denc + p * stride_p + (int64_t)(2 * level) * stride_k;
    const float gf[2] = {gptr[0], gptr[stride_k]};
    const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    float lx = 0.0f, ly = 0.0f, lz = 0.0f;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      auto q = [&](int k) { return f == 0 ? v[k].x : v[k].y; };
      const float yc_zc = q(7) * wx + q(6) * ux, yf_zc = q(5) * wx + q(4) * ux;
      const float yf_zf = q(1) * wx + q(0) * ux, yc_zf = q(3) * wx + q(2) * ux;
      const float zc = yc_zc * wy + yf_zc * uy, zf = yc_zf * wy + yf_zf * uy;
      const float g = gf[f];
      lz += g * (zc - zf);
      const float g_zc = g * wz, g_zf = g * uz;
      ly += g_zc * (yc_zc - yf_zc) + g_zf * (yc_zf - yf_zf);
      const float g_yczc = g_zc * wy, g_yfzc = g_zc * uy, g_yczf = g_zf * wy, g_yfzf = g_zf * uy;
      lx += g_yczc * (q(7) - q(6)) + g_yfzc * (q(5) - q(4)) + g_yfzf * (q(1) - q(0)) + g_yczf * (q(3) - q(2));
    }
    gx += lx * scale;
    gy += ly * scale;
    gz += lz * scale;
  }
  // back through `positions * selector`, the affine map and the contraction
  gx *= sel;
  gy *= sel;
  gz *= sel;
  if (transform == NSAMD_XFORM_CONTRACT) {
    gx /= 4.0f;
    gy /= 4.0f;
    gz /= 4.0f;
    contract_linf_bwd(rx, ry, rz, gx, gy, gz);
  } else if (transform == NSAMD_XFORM_AABB) {
    gx /= (box.hi[0] - box.lo[0]);
    gy /= (box.hi[1] - box.lo[1]);
    gz /= (box.hi[2] - box.lo[2]);
  }
  dpos[3 * p + 0] = gx;
  dpos[3 * p + 1] = gy;
  dpos[3 * p + 2] = gz;
}

__global__ void sh4_kernel(const float* __restrict__ dirs, int64_t M, float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const float x = dirs[3 * p], y = dirs[3 * p + 1], z = dirs[3 * p + 2];
  float c[16];
  sh4_components(x, y, z, c);
  float4* o = reinterpret_cast<float4*>(out + 16 * p);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = make_float4(c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]);
}

__global__ void contract_kernel(const float* __restrict__ in, int64_t M, float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  float x = in[3 * p], y = in[3 * p + 1], z = in[3 * p + 2];
  contract_linf(x, y, z);
  out[3 * p] = x;
  out[3 * p + 1] = y;
  out[3 * p + 2] = z;
}

static int check_points(const nsamd_points& P, int64_t M) {
  if (M < 0) return NSAMD_ERR_INVALID_ARG;
  if (P.positions == nullptr) {
    if (P.origins == nullptr || P.directions == nullptr || P.t_bins == nullptr || P.samples_per_ray <= 0)
      return NSAMD_ERR_INVALID_ARG;
    if (M % P.samples_per_ray != 0) return NSAMD_ERR_INVALID_ARG;
  }
  return NSAMD_OK;
}

static int check_grid(const nsamd_grid& g) {
  if (g.num_levels <= 0 || g.num_levels > NSAMD_MAX_LEVELS) return NSAMD_ERR_UNSUPPORTED;
  if (g.log2_table_size < 1 || g.log2_table_size > 28) return NSAMD_ERR_UNSUPPORTED;
  return NSAMD_OK;
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_hashgrid_encode_fwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                         const float* table, nsamd_grid grid, float* enc, int64_t stride_p,
                                         int64_t stride_k, float* selector, nsamd_stream_t stream) {
  if (M == 0) return NSAMD_OK;  // empty input: nothing to launch (empty tensors carry NULL data pointers)
  int st = check_points(pts, M);
  if (st) return st;
  st = check_grid(grid);
  if (st) return st;
  NSAMD_REQUIRE(table != nullptr && enc != nullptr);
  NSAMD_REQUIRE(transform >= 0 && transform <= 2);
  if (M == 0) return NSAMD_OK;
  const int64_t nb = (M + kHashBlock - 1) / kHashBlock;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  dim3 g((unsigned)nb, (unsigned)grid.num_levels);
  hash_encode_fwd_kernel<<<g, kHashBlock, 0, (hipStream_t)stream>>>(
      pts, M, transform, aabb, reinterpret_cast<const float2*>(table), grid, enc, stride_p, stride_k, selector);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_hashgrid_encode_bwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                         const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                         int64_t stride_k, float* dtable, float* dpositions,
                                         nsamd_stream_t stream) {
  if (M == 0) return NSAMD_OK;
  int st = check_points(pts, M);
  if (st) return st;
  st = check_grid(grid);
  if (st) return st;
  NSAMD_REQUIRE(denc != nullptr);
  NSAMD_REQUIRE(transform >= 0 && transform <= 2);
  NSAMD_REQUIRE(dtable != nullptr || dpositions != nullptr);
  if (M == 0) return NSAMD_OK;
  const int64_t nb = (M + kHashBlock - 1) / kHashBlock;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  if (dtable != nullptr) {
    dim3 g((unsigned)nb, (unsigned)grid.num_levels);
    hash_encode_bwd_table_kernel<<<g, kHashBlock, 0, (hipStream_t)stream>>>(pts, M, transform, aabb, grid, denc,
                                                                             stride_p, stride_k, dtable);
    NSAMD_CHECK_LAUNCH();
  }
  if (dpositions != nullptr) {
    NSAMD_REQUIRE(table != nullptr);
    hash_encode_bwd_pos_kernel<<<(unsigned)nb, kHashBlock, 0, (hipStream_t)stream>>>(
        pts, M, transform, aabb, reinterpret_cast<const float2*>(table), grid, denc, stride_p, stride_k,
        dpositions);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}

extern "C" int nsamd_sh4_encode(const float* dirs, int64_t M, float* out, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && (M == 0 || (dirs != nullptr && out != nullptr)));
  if (M == 0) return NSAMD_OK;
  sh4_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(dirs, M, out);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_contract_linf(const float* x, int64_t M, float* out, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && (M == 0 || (x != nullptr && out != nullptr)));
  if (M == 0) return NSAMD_OK;
  contract_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, M, out);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
