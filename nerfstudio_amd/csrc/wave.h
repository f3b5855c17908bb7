// Wave-level (64 lanes) scans and broadcasts of doubles on the DPP path of gfx9-family hardware.
// The generic __shfl_up / __shfl of a double compile to two ds_bpermute_b32 each — an LDS-crossbar round trip of ~100 clocks
// — and the per-ray kernels (one wavefront per ray) are chains of such scans: a 6-step Hillis-Steele scan was ~1.5 k clocks.
// DPP moves run at VALU rate: row_shr 1/2/4/8 scan the four rows of 16 lanes, row_bcast:15 / row_bcast:31 carry the row
// totals across (the sequence the AMDGPU backend itself emits for wave64 scans on GFX9).
// Association of the additions differs from Hillis-Steele; for the fp32-valued summands of this library the double partial
// sums are exact (sampler.hip header), so the fp32-rounded results are the same numbers.
#pragma once
#include <hip/hip_runtime.h>

namespace nsamd {

// Index of this wavefront inside its workgroup. NSAMD_SCALAR_RAY=1 at build time makes it a SCALAR (wave-uniform by
// construction, which the compiler cannot prove of threadIdx.x >> 6): everything derived from it — the wave's ray, its row
// pointers, the ray's near / far / jitter — then lives in scalar registers and is fetched through the scalar cache. Measured on
// MI355X (profiles/r05_s5_ab_scalar_ray.txt): nothing — every per-ray launch within 1 % either way, the step 0.698 against
// 0.692 ms — so the plain expression is the default: the scalar cache is not coherent with a wave's own vector stores inside
// one launch, which a kernel that reads back what it wrote (csrc/fused_rays.hip) would have to keep in mind for no gain.
#ifndef NSAMD_SCALAR_RAY
#define NSAMD_SCALAR_RAY 0
#endif
__device__ __forceinline__ int wave_index() {
#if NSAMD_SCALAR_RAY
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
  return (int)(threadIdx.x >> 6);
#endif
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  int lo = (int)(unsigned)(b & 0xffffffffull), hi = (int)(unsigned)(b >> 32);
  // old = 0, bound_ctrl: lanes without a source (row start, disabled rows) read +0.0
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_scan_inclusive_f64(double v) {
  v = v + dpp_move_f64<0x111, 0xf>(v);  // row_shr:1
  v = v + dpp_move_f64<0x112, 0xf>(v);  // row_shr:2
  v = v + dpp_move_f64<0x114, 0xf>(v);  // row_shr:4
  v = v + dpp_move_f64<0x118, 0xf>(v);  // row_shr:8   -> inclusive scan inside every row of 16
  v = v + dpp_move_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = v + dpp_move_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// value of the previous lane (lane 0 reads +0.0): wave_shr:1
__device__ __forceinline__ double wave_shift_up1_f64(double v) { return dpp_move_f64<0x138, 0xf>(v); }

// value of lane `LANE`, wave-uniform (v_readlane_b32 x 2)
template <int LANE>
__device__ __forceinline__ double wave_read_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffull), LANE);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), LANE);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}

// Value of the neighbouring lane of the pair (2i, 2i + 1): DPP quad_perm [1, 0, 3, 2] — VALU rate, no LDS. All four lanes of
// the quad must be active.
__device__ __forceinline__ uint32_t pair_swap_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
__device__ __forceinline__ float pair_swap_f32(float v) { return __uint_as_float(pair_swap_u32(__float_as_uint(v))); }

// Sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), valid in the row's lane 0 (and, up to the association, in every
// lane): the xor-butterfly's additions for that lane — partners 1, 2, 4, 8 — as quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_ror:12, row_ror:8 at VALU rate. (`__shfl_xor` compiles to ds_bpermute_b32: an LDS-crossbar round trip per step; the field
// backward's appearance-gradient rows took 32 of them per tile, four deep.) Same bits in lane 0 as the butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_move_f32(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum_lane0(float v) {
  v += dpp_move_f32<0xB1>(v);
  v += dpp_move_f32<0x4E>(v);
  v += dpp_move_f32<0x12C>(v);  // row_ror:12: lane i reads lane (i + 4) % 16 — lane 0 its butterfly partner 4, lane 8 its partner 12
  v += dpp_move_f32<0x128>(v);  // row_ror:8:  lane 0 reads lane 8
  return v;
}

}  // namespace nsamd
