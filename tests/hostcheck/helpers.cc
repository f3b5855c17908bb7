// TEST INFRASTRUCTURE (tests/test_kernel_helpers_cpu.py): compiles the per-point arithmetic the HIP kernels share
// (nerfstudio_amd/csrc/common.h: NSAMD_HD host+device functions; scatter.h: the fixed-point accumulation) with the HOST
// compiler, so that `pytest -m "not gpu"` can pin those functions to the reference's known answers and to the oracle
// without a GPU. Nothing in the product loads this library; the kernels themselves run these functions on the device.
#include <math.h>
#include <stdint.h>

// scatter.h is device-only source: give the host compiler the three names it needs (no kernels are compiled here)
#define __device__
#define __forceinline__ inline
struct uint4 { unsigned x, y, z, w; };
typedef void* hipStream_t;
#include "../../nerfstudio_amd/csrc/scatter.h"
#include "../../nerfstudio_amd/csrc/camera.h"

using namespace nsamd;

extern "C" {

void hc_hash_corners(const float* xyz, int64_t n, float scale, int log2_table_size, int64_t* out /* [n,8] */) {
  const uint32_t mask = (1u << log2_table_size) - 1u;
  for (int64_t p = 0; p < n; ++p) {
    const Cell c = locate_cell(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2], scale);
    for (int k = 0; k < 8; ++k) out[8 * p + k] = corner_index(c, k, mask);
  }
}

void hc_cell_weights(const float* xyz, int64_t n, float scale, float* w /* [n,3] */) {
  for (int64_t p = 0; p < n; ++p) {
    const Cell c = locate_cell(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2], scale);
    for (int a = 0; a < 3; ++a) w[3 * p + a] = c.w[a];
  }
}

uint32_t hc_hash_fn(int32_t ix, int32_t iy, int32_t iz, int log2_table_size) {
  return hash_corner(ix, iy, iz, (1u << log2_table_size) - 1u);
}

void hc_contract(float* xyz, int64_t n) {
  for (int64_t p = 0; p < n; ++p) contract_linf(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
}

void hc_contract_bwd(const float* xyz, float* g, int64_t n) {
  for (int64_t p = 0; p < n; ++p)
    contract_linf_bwd(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2], g[3 * p], g[3 * p + 1], g[3 * p + 2]);
}

// positions -> grid input in [0,1] and the selector; transform: nsamd_transform of include/nsamd.h (0 none, 1 contraction, 2 aabb)
void hc_normalise(float* xyz, int64_t n, int transform, const float* lo, const float* hi, float* sel) {
  nsamd_aabb box;
  for (int a = 0; a < 3; ++a) {
    box.lo[a] = lo[a];
    box.hi[a] = hi[a];
  }
  for (int64_t p = 0; p < n; ++p) sel[p] = normalise_position(transform, box, xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
}

void hc_sh4(const float* d, int64_t n, float* out /* [n,16] */) {
  for (int64_t p = 0; p < n; ++p) sh4_components(d[3 * p], d[3 * p + 1], d[3 * p + 2], out + 16 * p);
}

void hc_spacing(const float* x, int64_t n, float* fwd, float* inv) {
  for (int64_t i = 0; i < n; ++i) {
    fwd[i] = spacing_fn(x[i]);
    inv[i] = spacing_fn_inv(x[i]);
  }
}

float hc_spacing_to_euclidean(int mode, float s, float s_near, float s_far) {
  return spacing_to_euclidean_mode(mode, s, s_near, s_far);
}

// Frustums.get_positions through the ray form of nsamd_points
void hc_positions(const float* origins, const float* directions, const float* t_bins, int64_t rays, int64_t S,
                  float* out /* [rays*S,3] */) {
  nsamd_points P;
  P.positions = nullptr;
  P.origins = origins;
  P.directions = directions;
  P.t_bins = t_bins;
  P.samples_per_ray = S;
  for (int64_t p = 0; p < rays * S; ++p) load_position(P, p, out[3 * p], out[3 * p + 1], out[3 * p + 2]);
}

// ... and through the two "burst" forms the hash forward and the scatter's route kernels use (same arithmetic, loads laid
// out differently): variant 1 = load_position_burst, 2 = load_positions_burst<4> over 4 consecutive points
void hc_positions_burst(const float* origins, const float* directions, const float* t_bins, const float* positions,
                        int64_t rays, int64_t S, int variant, float* out /* [rays*S,3] */) {
  nsamd_points P;
  P.positions = positions;
  P.origins = origins;
  P.directions = directions;
  P.t_bins = t_bins;
  P.samples_per_ray = S;
  const int64_t M = rays * S;
  if (variant == 1) {
    for (int64_t p = 0; p < M; ++p) load_position_burst(P, p, out[3 * p], out[3 * p + 1], out[3 * p + 2]);
    return;
  }
  for (int64_t p0 = 0; p0 < M; p0 += 4) {
    int64_t p[4];
    float x[4], y[4], z[4];
    for (int k = 0; k < 4; ++k) p[k] = p0 + k < M ? p0 + k : M - 1;  // (clamped, as the kernels do)
    load_positions_burst<4>(P, p, x, y, z);
    for (int k = 0; k < 4 && p0 + k < M; ++k) {
      out[3 * (p0 + k)] = x[k];
      out[3 * (p0 + k) + 1] = y[k];
      out[3 * (p0 + k) + 2] = z[k];
    }
  }
}

void hc_nan_to_num(float* v, int64_t n, float nan_value) {
  for (int64_t i = 0; i < n; ++i) v[i] = nan_to_num(v[i], nan_value);
}

// ---- fixed-point accumulation of the table scatter ----
void hc_fixed_scale(uint32_t max_bits, int headroom, int* k, int* empty, int* bad) {
  const FixedScale s = fixed_scale(max_bits, headroom);
  *k = s.k;
  *empty = s.empty;
  *bad = s.bad;
}

void hc_to_fixed(const float* v, int64_t n, int k, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)to_fixed(v[i], k);
}

// sum of the fixed-point images of v in the given order (modulo 2^64, as the LDS atomics add) -> float
float hc_fixed_sum(const float* v, const int64_t* order, int64_t n, int k) {
  unsigned long long acc = 0ull;
  for (int64_t i = 0; i < n; ++i) acc += to_fixed(v[order[i]], k);
  return from_fixed(acc, k);
}

// ---- camera-pose corrections (csrc/camera.h) ----
void hc_cam_exp_map(int mode, const float* pose, int64_t n, float* out /* [n,3,4] = [R|t] */) {
  for (int64_t c = 0; c < n; ++c) {
    float R[9], t[3];
    cam_exp_map(mode, pose + 6 * c, R, t);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) out[12 * c + 4 * i + j] = R[3 * i + j];
      out[12 * c + 4 * i + 3] = t[i];
    }
  }
}

// upstream [n,3,4] = dL/d[R|t] -> dpose [n,6]; reg != 0: the regulariser's gradient is added (penalties / n as the mean does)
void hc_cam_exp_map_bwd(int mode, const float* pose, const float* upstream, int64_t n, float trans_pen, float rot_pen,
                        int reg, float* dpose) {
  for (int64_t c = 0; c < n; ++c) {
    double G[9], g[3], dp[6], dr[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) G[3 * i + j] = upstream[12 * c + 4 * i + j];
      g[i] = upstream[12 * c + 4 * i + 3];
    }
    cam_exp_map_bwd(mode, pose + 6 * c, G, g, dp);
    if (reg) cam_reg_bwd(pose + 6 * c, (double)trans_pen / n, (double)rot_pen / n, dr);
    for (int k = 0; k < 6; ++k) dpose[6 * c + k] = (float)dp[k] + (float)dr[k];
  }
}

}  // extern "C"
