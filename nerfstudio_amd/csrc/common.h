// Shared helpers for the nsamd HIP kernels (gfx950 only).
// Per-point / per-ray arithmetic lives in NSAMD_HD (host + device) inline functions shared by all kernels; the host
// compiler only sees them when it checks the header (tests/test_abi.py). The product never runs them on the CPU.
#pragma once

#include <stdint.h>
#include <math.h>

#include "../../include/nsamd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NSAMD_HD __host__ __device__ __forceinline__
#define NSAMD_D __device__ __forceinline__
#else
#define NSAMD_HD inline
#define NSAMD_D inline
#endif

#if defined(__HIPCC__)
#define NSAMD_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return NSAMD_ERR_LAUNCH;            \
  } while (0)
#endif

#if defined(__HIPCC__)
// A pointer the compiler cannot trace back to a kernel argument (read from LDS, from a struct in memory, passed to a
// non-inlined function) is GENERIC, and its accesses compile to FLAT instructions. Those count in vmcnt AND lgkmcnt, and a
// wave's next `s_waitcnt lgkmcnt` — every LDS operand fetch of the MFMA loops — then waits for the flat access to come back
// from memory: a fire-and-forget record store turns into a full L2 round trip on the matrix pipeline's critical path (read off
// the ISA of the field backward that emits scatter records). Device buffers are global memory: say so.
#define NSAMD_GLOBAL_AS __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const NSAMD_GLOBAL_AS T* global_ptr(const T* p) {
  return (const NSAMD_GLOBAL_AS T*)p;
}
template <class T>
__device__ __forceinline__ NSAMD_GLOBAL_AS T* global_ptr(T* p) {
  return (NSAMD_GLOBAL_AS T*)p;
}
#endif

// compiler barrier between two loads that must not be merged into one wider load (device code; nothing on the host)
#if defined(__HIP_DEVICE_COMPILE__)
#define NSAMD_KEEP_LOADS_APART() asm volatile("" ::: "memory")
#else
#define NSAMD_KEEP_LOADS_APART() do {} while (0)
#endif

#define NSAMD_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return NSAMD_ERR_INVALID_ARG; \
  } while (0)

// Timing probes (scripts/probe_*_clocks.py build a probe library with -DNSAMD_PROBE_CLOCKS; never part of libnsamd.so):
// lane 0 of every wave stamps the shader clock into [wave of the grid][64 slots] of a buffer set per translation unit.
#ifdef NSAMD_PROBE_CLOCKS
#define NSAMD_PROBE_DEFINE(tag)                                                                          \
  static __device__ long long* g_probe_clocks = nullptr;                                                 \
  extern "C" int nsamd_probe_set_clocks_##tag(long long* buffer) {                                       \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_probe_clocks), &buffer, sizeof(buffer)) == hipSuccess ? 0 : -3; \
  }
#define PROBE_STAMP(unused, slot)                                                                        \
  do {                                                                                                   \
    if (g_probe_clocks != nullptr && (threadIdx.x & 63) == 0 && (slot) < 64)                             \
      g_probe_clocks[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 + \
                     (slot)] = clock64();                                                                \
  } while (0)
#else
#define NSAMD_PROBE_DEFINE(tag)
#define PROBE_STAMP(unused, slot) do {} while (0)
#endif

namespace nsamd {

constexpr uint32_t kPrimeY = 2654435761u;  // encodings.py:410
constexpr uint32_t kPrimeZ = 805459861u;   // encodings.py:410

// ray of point p for S samples per ray: a 32-bit division whenever both fit (always, below 4 G points) — a 64-bit divide is
// ~60 vector instructions. (Round 5 measured what that buys: nothing — 919 -> 860 vector instructions per 64 points of the
// proposal field, and 718 with the two features of a corner blended as one packed operation, leave its launch at 40.5 us and
// the hash forward at 89 us on the same box: both are bound by their L1 line lookups, not by instruction issue. The division
// stays, the packed blends — three more registers, 27 more for 8-level grids — went again; profiles/r05_s4_ab_valu_diet.txt.)
NSAMD_HD int64_t point_ray(int64_t p, int64_t S) {
  return (((uint64_t)p | (uint64_t)S) >> 32) ? p / S : (int64_t)((uint32_t)p / (uint32_t)S);
}

// ---- position of sample p (Frustums.get_positions, cameras/rays.py:50-59) -------------------------------------
NSAMD_HD void load_position(const nsamd_points& P, int64_t p, float& x, float& y, float& z) {
  if (P.positions != nullptr) {
    x = P.positions[3 * p + 0];
    y = P.positions[3 * p + 1];
    z = P.positions[3 * p + 2];
  } else {
    const int64_t S = P.samples_per_ray;
    const int64_t ray = point_ray(p, S);
    const int64_t s = p - ray * S;
    const float* tb = P.t_bins + ray * (S + 1) + s;
    const float span = tb[0] + tb[1];  // starts + ends
    const float* o = P.origins + 3 * ray;
    const float* d = P.directions + 3 * ray;
    x = o[0] + d[0] * span / 2.0f;
    y = o[1] + d[1] * span / 2.0f;
    z = o[2] + d[2] * span / 2.0f;
  }
}

// The same position with its eight loads issued back to back as single dwords (compiler barriers keep them from being merged
// and re-ordered): ONE memory round trip. In the order of `load_position` — bin edges, their sum, then origin and direction —
// the compiler waits for the edges before it issues the other loads (two round trips in a row), and merged into dwordx3 the
// triples land in register tuples that are re-packed right behind the load (another wait). Worth it where the kernel is a chain
// of latencies with few memory instructions per lane (the main-grid hash forward: 93.0 -> 89.3 us); NOT where the load / gather
// instructions themselves are the bound (the fused proposal field, 40 gathers per lane: 43 -> 57 us with five more loads).
NSAMD_HD void load_position_burst(const nsamd_points& P, int64_t p, float& x, float& y, float& z) {
  if (P.positions != nullptr) {
    load_position(P, p, x, y, z);
    return;
  }
  const int64_t S = P.samples_per_ray;
  const int64_t ray = point_ray(p, S);
  const int64_t s = p - ray * S;
  const float* tb = P.t_bins + ray * (S + 1) + s;
  const float* o = P.origins + 3 * ray;
  const float* d = P.directions + 3 * ray;
  const float o0 = o[0];
  NSAMD_KEEP_LOADS_APART();
  const float o1 = o[1];
  NSAMD_KEEP_LOADS_APART();
  const float o2 = o[2];
  NSAMD_KEEP_LOADS_APART();
  const float d0 = d[0];
  NSAMD_KEEP_LOADS_APART();
  const float d1 = d[1];
  NSAMD_KEEP_LOADS_APART();
  const float d2 = d[2];
  NSAMD_KEEP_LOADS_APART();
  const float t0 = tb[0];
  NSAMD_KEEP_LOADS_APART();
  const float t1 = tb[1];
  const float span = t0 + t1;  // starts + ends
  x = o0 + d0 * span / 2.0f;
  y = o1 + d1 * span / 2.0f;
  z = o2 + d2 * span / 2.0f;
}

// N positions of one lane with ONE branch on the layout around all of their loads (a branch per position is a wait per
// position: the compiler closes each with s_waitcnt vmcnt(0)).
template <int N>
NSAMD_HD void load_positions_burst(const nsamd_points& P, const int64_t (&p)[N], float (&x)[N], float (&y)[N], float (&z)[N]) {
  if (P.positions != nullptr) {
    for (int k = 0; k < N; ++k) {
      x[k] = P.positions[3 * p[k] + 0];
      y[k] = P.positions[3 * p[k] + 1];
      z[k] = P.positions[3 * p[k] + 2];
    }
    return;
  }
  const int64_t S = P.samples_per_ray;
  float o0[N], o1[N], o2[N], d0[N], d1[N], d2[N], t0[N], t1[N];
  for (int k = 0; k < N; ++k) {
    const int64_t ray = point_ray(p[k], S);
    const int64_t s = p[k] - ray * S;
    const float* tb = P.t_bins + ray * (S + 1) + s;
    const float* o = P.origins + 3 * ray;
    const float* d = P.directions + 3 * ray;
    o0[k] = o[0];
    NSAMD_KEEP_LOADS_APART();
    o1[k] = o[1];
    NSAMD_KEEP_LOADS_APART();
    o2[k] = o[2];
    NSAMD_KEEP_LOADS_APART();
    d0[k] = d[0];
    NSAMD_KEEP_LOADS_APART();
    d1[k] = d[1];
    NSAMD_KEEP_LOADS_APART();
    d2[k] = d[2];
    NSAMD_KEEP_LOADS_APART();
    t0[k] = tb[0];
    NSAMD_KEEP_LOADS_APART();
    t1[k] = tb[1];
    NSAMD_KEEP_LOADS_APART();
  }
  for (int k = 0; k < N; ++k) {
    const float span = t0[k] + t1[k];  // starts + ends
    x[k] = o0[k] + d0[k] * span / 2.0f;
    y[k] = o1[k] + d1[k] * span / 2.0f;
    z[k] = o2[k] + d2[k] * span / 2.0f;
  }
}

// ---- L-inf scene contraction (spatial_distortions.py:66-69) ---------------------------------------------------
NSAMD_HD void contract_linf(float& x, float& y, float& z) {
  const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  if (!(mag < 1.0f)) {  // torch.where(mag < 1, x, ...): NaN takes the second branch
    const float a = 2.0f - (1.0f / mag);
    x = a * (x / mag);
    y = a * (y / mag);
    z = a * (z / mag);
  }
}

// Backward of contract_linf as autograd differentiates `(2 - 1/mag) * (x / mag)` with mag = ||x||_inf
// (linalg_vector_norm backward shares the gradient equally between tied maxima).
NSAMD_HD void contract_linf_bwd(float x, float y, float z, float& gx, float& gy, float& gz) {
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  const float mag = fmaxf(ax, fmaxf(ay, az));
  if (mag < 1.0f) return;
  const float a = 2.0f - (1.0f / mag);
  const float inv = 1.0f / mag;
  // y_i = a * b_i, b_i = x_i / mag
  const float g_a = gx * (x * inv) + gy * (y * inv) + gz * (z * inv);
  float g_mag = g_a * (inv * inv);  // d a / d mag = 1/mag^2
  // d b_i / d mag = -x_i / mag^2
  g_mag -= a * (gx * x + gy * y + gz * z) * (inv * inv);
  const float tx = (ax == mag) ? 1.0f : 0.0f, ty = (ay == mag) ? 1.0f : 0.0f, tz = (az == mag) ? 1.0f : 0.0f;
  const float cnt = tx + ty + tz;
  const float sx = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
  const float sy = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f);
  const float sz = (z > 0.0f) ? 1.0f : ((z < 0.0f) ? -1.0f : 0.0f);
  gx = gx * a * inv + g_mag * sx * tx / cnt;
  gy = gy * a * inv + g_mag * sy * ty / cnt;
  gz = gz * a * inv + g_mag * sz * tz / cnt;
}

// ---- raw position -> hash-grid input in [0,1] + selector (density_fields.py:95-103) ---------------------------
NSAMD_HD float normalise_position(int transform, const nsamd_aabb& box, float& x, float& y, float& z) {
  if (transform == NSAMD_XFORM_NONE) return 1.0f;
  if (transform == NSAMD_XFORM_CONTRACT) {
    contract_linf(x, y, z);
    x = (x + 2.0f) / 4.0f;
    y = (y + 2.0f) / 4.0f;
    z = (z + 2.0f) / 4.0f;
  } else {
    x = (x - box.lo[0]) / (box.hi[0] - box.lo[0]);
    y = (y - box.lo[1]) / (box.hi[1] - box.lo[1]);
    z = (z - box.lo[2]) / (box.hi[2] - box.lo[2]);
  }
  const bool inside = (x > 0.0f) && (x < 1.0f) && (y > 0.0f) && (y < 1.0f) && (z > 0.0f) && (z < 1.0f);
  const float sel = inside ? 1.0f : 0.0f;
  x *= sel;  // positions * selector[..., None]
  y *= sel;
  z *= sel;
  return sel;
}

// ---- spatial hash (HashEncoding.hash_fn, encodings.py:398-415) in wrap-around uint32 --------------------------
NSAMD_HD uint32_t hash_corner(int32_t ix, int32_t iy, int32_t iz, uint32_t mask) {
  return ((uint32_t)ix ^ ((uint32_t)iy * kPrimeY) ^ ((uint32_t)iz * kPrimeZ)) & mask;
}

// Per-level cell data of one point: integer floor/ceil corners and the ceil-corner blend weights.
struct Cell {
  int32_t lo[3];
  int32_t hi[3];
  float w[3];
};

NSAMD_HD Cell locate_cell(float x, float y, float z, float scale) {
  Cell c;
  const float sx = x * scale, sy = y * scale, sz = z * scale;
  const float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  c.lo[0] = (int32_t)fx;
  c.lo[1] = (int32_t)fy;
  c.lo[2] = (int32_t)fz;
  c.hi[0] = (int32_t)ceilf(sx);
  c.hi[1] = (int32_t)ceilf(sy);
  c.hi[2] = (int32_t)ceilf(sz);
  c.w[0] = sx - fx;
  c.w[1] = sy - fy;
  c.w[2] = sz - fz;
  return c;
}

// Corner order used throughout: bit0 = x is ceil, bit1 = y is ceil, bit2 = z is ceil.
NSAMD_HD uint32_t corner_index(const Cell& c, int corner, uint32_t mask) {
  return hash_corner((corner & 1) ? c.hi[0] : c.lo[0], (corner & 2) ? c.hi[1] : c.lo[1],
                     (corner & 4) ? c.hi[2] : c.lo[2], mask);
}

// ---- piecewise spacing function (ray_samplers.py:244-245) -----------------------------------------------------
NSAMD_HD float spacing_fn(float x) { return (x < 1.0f) ? (x / 2.0f) : (1.0f - 1.0f / (2.0f * x)); }
NSAMD_HD float spacing_fn_inv(float x) { return (x < 0.5f) ? (2.0f * x) : (1.0f / (2.0f - 2.0f * x)); }
// closure of ray_samplers.py:115-116
NSAMD_HD float spacing_to_euclidean(float s, float s_near, float s_far) {
  return spacing_fn_inv(s * s_far + (1.0f - s) * s_near);
}
// spacing 0 = UniformLinDispPiecewiseSampler (ray_samplers.py:244-245), 1 = UniformSampler (identity, :131-155)
NSAMD_HD float spacing_fn_mode(int spacing, float x) { return spacing == 1 ? x : spacing_fn(x); }
NSAMD_HD float spacing_to_euclidean_mode(int spacing, float s, float s_near, float s_far) {
  const float v = s * s_far + (1.0f - s) * s_near;
  return spacing == 1 ? v : spacing_fn_inv(v);
}

// ---- real spherical harmonics, 4 levels (utils/spherical_harmonics.py:24-93), same association order -----------
NSAMD_HD void sh4_components(float x, float y, float z, float* c) {
  const float xx = x * x, yy = y * y, zz = z * z;
  c[0] = 0.28209479177387814f;
  c[1] = 0.4886025119029199f * y;
  c[2] = 0.4886025119029199f * z;
  c[3] = 0.4886025119029199f * x;
  c[4] = 1.0925484305920792f * x * y;
  c[5] = 1.0925484305920792f * y * z;
  c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  c[7] = 1.0925484305920792f * x * z;
  c[8] = 0.5462742152960396f * (xx - yy);
  c[9] = 0.5900435899266435f * y * (3.0f * xx - yy);
  c[10] = 2.890611442640554f * x * y * z;
  c[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
  c[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
  c[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
  c[14] = 1.445305721320277f * z * (xx - yy);
  c[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

// torch.nan_to_num defaults: nan -> 0 (or `nan`), +inf -> FLT_MAX, -inf -> -FLT_MAX
NSAMD_HD float nan_to_num(float v, float nan_value = 0.0f) {
  if (v != v) return nan_value;
  if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
  if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
  return v;
}

}  // namespace nsamd
