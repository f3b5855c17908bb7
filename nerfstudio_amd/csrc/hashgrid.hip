// Multiresolution hash encoding for gfx950, torch-path semantics of the reference
// (HashEncoding.pytorch_fwd, /root/reference/nerfstudio/field_components/encodings.py:417-458).
//
// Forward mapping: grid = (ceil(M/256), L / levels-per-thread). For the nerfacto main grid one thread = one (point,
// level) and blockIdx.y is the LEVEL, so the blocks resident at any moment sweep one or two levels of the table
// (2^log2_T * 8 B each; 4 MiB = one XCD's L2) instead of all L — the gathers are L2 hits instead of Infinity-Cache/HBM
// trips; for small tables (all levels of a proposal grid fit one L2 together) a thread takes 4 levels: the position is
// computed once and 32 gathers are in flight. Consecutive lanes are consecutive samples of one ray, i.e. spatial
// neighbours: on the coarse levels most of a wavefront reads the same few 128-B lines. The feature-major output
// (stride_p = 1) makes the 8-B-per-point result store one coalesced 256-B row per wavefront and feature; the [M, 2L]
// row-major layout of the stand-alone Encoding API is the strided variant.
//
// HBM-bound integer/gather work, no MFMA. Algorithmic bytes: 8 corners x 8 B per point and level (fwd),
// 8 corners x 16 B read-modify-write (bwd).
//
// Backward = scatter-add into the table gradient. Measured on MI355X (scripts/probe_scatter*.py, profiles/): fp32
// global atomics retire at a flat ~20 G lane-ops/s chip-wide whatever the locality (they are serviced memory-side,
// past the per-XCD L2s; contended coarse levels drop to 6 G/s) — 15x below the gather rate, 2.9 ms for the nerfacto
// main table. So the big scatter uses NO global atomics on its normal path ("binned" scatter, needs scratch from the
// caller): the table gradient is partitioned into (level, up-to-16K-entry) tiles; PASS 1 derives every corner update
// once and appends a 16-B record to the queue of the tile it falls into — a workgroup-local counting sort in LDS, one
// returning global atomic per (workgroup, non-empty tile) to reserve queue space, one dwordx4 store per record — with
// two kernels chosen per level (fine levels: x-pair records, 4 levels per thread; coarse levels: run merging + a
// workgroup-wide combining table); PASS 2 runs one workgroup per tile that accumulates its queue in an LDS tile
// (CAS on the float pair; ds_add_f32 with divergent addresses retires only 0.33 lane-ops/clk/CU) and adds — or, for
// the write-only entry point, stores — the tile with plain coalesced float4 accesses. Queues are sized 2x the
// uniform-hash expectation; what does not fit goes out as direct atomics (accumulating call) or through a deferred
// list applied after pass 2 (write-only call), so the result never depends on sizing. Fallbacks: without scratch one
// workgroup OWNS a tile and scans all sample points (hash_encode_bwd_sliced_kernel, 32x redundant hashing, 0.70 ms for
// the main table); tiny problems keep the direct-atomic kernel. DESIGN.md 4.1 has the measurements behind each choice.
#include <stdlib.h>

#include "common.h"
#include "scatter.h"
#include "wave.h"

namespace nsamd {

constexpr int kHashBlock = 256;

// kLevels levels per thread: the position (ray fetch + contraction) is computed once and 8 * kLevels gathers are in
// flight per lane.
template <int kLevels>
__global__ __launch_bounds__(kHashBlock) void hash_encode_fwd_kernel(nsamd_points P, int64_t M, int transform,
                                                                     nsamd_aabb box,
                                                                     const float2* __restrict__ table,
                                                                     nsamd_grid grid, float* __restrict__ enc,
                                                                     int64_t stride_p, int64_t stride_k,
                                                                     float* __restrict__ selector) {
  const int level0 = blockIdx.y * kLevels;
  const int64_t p = (int64_t)blockIdx.x * kHashBlock + threadIdx.x;
  if (p >= M) return;
  float x, y, z;
  load_position(P, p, x, y, z);
  const float sel = normalise_position(transform, box, x, y, z);
  if (level0 == 0 && selector != nullptr) selector[p] = sel;
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  float2 v[kLevels][8];
  float w[kLevels][3];
#pragma unroll
  for (int i = 0; i < kLevels; ++i) {
    const int level = level0 + i;
    if (level >= grid.num_levels) break;
    const Cell c = locate_cell(x, y, z, grid.scalings[level]);
    w[i][0] = c.w[0]; w[i][1] = c.w[1]; w[i][2] = c.w[2];
    const float2* __restrict__ tl = table + ((size_t)level << grid.log2_table_size);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[i][k] = tl[corner_index(c, k, mask)];
  }
#pragma unroll
  for (int i = 0; i < kLevels; ++i) {
    const int level = level0 + i;
    if (level >= grid.num_levels) break;
    const float wx = w[i][0], wy = w[i][1], wz = w[i][2];
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    float r[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      auto g = [&](int k) { return f == 0 ? v[i][k].x : v[i][k].y; };
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
}

// One level per thread, two switches measured against the kernel above (NSAMD_HASH_FWD_MODE bits 1 / 2):
//  kPair: the two x-neighbours of a cell edge differ by lo ^ hi in their hashed index whatever y and z are; when that is 1
//         (lo even) or 0 (the point sits on a lattice plane) both entries lie in one aligned 16-B chunk, so ONE dwordx4 gather
//         serves the pair and the second, lane-masked gather only runs for odd lo — 6 instead of 8 L1 accesses per (point,
//         level) on average. The gathers are bound by L1 tag lookups (r02: 23.9 M accesses, 0.53 per clock per CU), not bytes.
//  kXcd:  1-D grid, block b runs on XCD b % 8 (observed placement, speed only): XCD x sweeps levels x, 15 - x, 16 + x, ... so a
//         level slice is filled into ONE L2 instead of all eight.
template <bool kPair, bool kXcd>
__global__ __launch_bounds__(kHashBlock) void hash_encode_fwd_v2_kernel(nsamd_points P, int64_t M, int transform,
                                                                        nsamd_aabb box, const float2* __restrict__ table,
                                                                        nsamd_grid grid, float* __restrict__ enc,
                                                                        int64_t stride_p, int64_t stride_k,
                                                                        float* __restrict__ selector, unsigned nb) {
  int level;
  int64_t pb;
  if (kXcd) {
    const unsigned xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const unsigned li = q / nb;
    pb = q - li * nb;
    level = (li & 1u) ? (int)(8u * li + 7u - xcd) : (int)(8u * li + xcd);
    if (level >= grid.num_levels) return;
  } else {
    level = blockIdx.y;
    pb = blockIdx.x;
  }
  const int64_t p = pb * kHashBlock + threadIdx.x;
  if (p >= M) return;
  float x, y, z;
  load_position_burst(P, p, x, y, z);
  const float sel = normalise_position(transform, box, x, y, z);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const Cell c = locate_cell(x, y, z, grid.scalings[level]);
  const float2* __restrict__ tl = table + ((size_t)level << grid.log2_table_size);
  float2 v0, v1, v2, v3, v4, v5, v6, v7;
  if (kPair) {
    const float4* __restrict__ tl4 = reinterpret_cast<const float4*>(tl);
    const uint32_t hy0 = (uint32_t)c.lo[1] * kPrimeY, hy1 = (uint32_t)c.hi[1] * kPrimeY;
    const uint32_t hz0 = (uint32_t)c.lo[2] * kPrimeZ, hz1 = (uint32_t)c.hi[2] * kPrimeZ;
    const uint32_t xl = (uint32_t)c.lo[0], xh = (uint32_t)c.hi[0];
    const uint32_t a0 = (xl ^ hy0 ^ hz0) & mask, a1 = (xl ^ hy1 ^ hz0) & mask, a2 = (xl ^ hy0 ^ hz1) & mask,
                   a3 = (xl ^ hy1 ^ hz1) & mask;
    const uint32_t b0 = (xh ^ hy0 ^ hz0) & mask, b1 = (xh ^ hy1 ^ hz0) & mask, b2 = (xh ^ hy0 ^ hz1) & mask,
                   b3 = (xh ^ hy1 ^ hz1) & mask;
    const float4 q0 = tl4[a0 >> 1], q1 = tl4[a1 >> 1], q2 = tl4[a2 >> 1], q3 = tl4[a3 >> 1];
    auto half = [](const float4& q, uint32_t i) { return (i & 1u) ? make_float2(q.z, q.w) : make_float2(q.x, q.y); };
    v0 = half(q0, a0); v2 = half(q1, a1); v4 = half(q2, a2); v6 = half(q3, a3);
    if ((((xl ^ xh) & mask) >> 1) == 0u) {
      v1 = half(q0, b0); v3 = half(q1, b1); v5 = half(q2, b2); v7 = half(q3, b3);
    } else {
      v1 = tl[b0]; v3 = tl[b1]; v5 = tl[b2]; v7 = tl[b3];
    }
  } else {
    v0 = tl[corner_index(c, 0, mask)]; v1 = tl[corner_index(c, 1, mask)]; v2 = tl[corner_index(c, 2, mask)];
    v3 = tl[corner_index(c, 3, mask)]; v4 = tl[corner_index(c, 4, mask)]; v5 = tl[corner_index(c, 5, mask)];
    v6 = tl[corner_index(c, 6, mask)]; v7 = tl[corner_index(c, 7, mask)];
  }
  const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
  const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
  float r[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    auto g = [&](const float2& a) { return f == 0 ? a.x : a.y; };
    const float yc_zc = g(v7) * wx + g(v6) * ux;  // blend order x, y, z exactly as encodings.py:446-456
    const float yf_zc = g(v5) * wx + g(v4) * ux;
    const float yf_zf = g(v1) * wx + g(v0) * ux;
    const float yc_zf = g(v3) * wx + g(v2) * ux;
    const float zc = yc_zc * wy + yf_zc * uy;
    const float zf = yc_zf * wy + yf_zf * uy;
    r[f] = zc * wz + zf * uz;
  }
  float* o = enc + p * stride_p + (int64_t)(2 * level) * stride_k;
  o[0] = r[0];
  o[stride_k] = r[1];
  // (behind the gathers: vector memory operations retire in order and stores count — in front of them the selector store
  //  would have to complete before the first gathered value may be used)
  if (level == 0 && selector != nullptr) selector[p] = sel;
}

// kLanePair (NSAMD_HASH_FWD_MODE bit 4; round 5). What a divergent gather costs on this part is the number of distinct 128-B
// LINES a wave instruction touches — ~0.47 lines per clock and CU whatever the access width, and lanes that share a line are
// free (scripts/probe_gather.hip, profiles/r05_s14_probe_gather.txt: pairs of lanes on one line 2 x, quads 4 x the lane rate;
// the forward at fine levels runs the address unit 91 % busy, profiles/r05_s13_hash_fwd_cache_counters.txt). The two
// x-neighbours of a cell edge sit in ONE line 15 times out of 16 (their indices differ by lo ^ hi = 2^(k+1) - 1, k = trailing
// ones of lo; a line holds 16 entries) but, fetched by the same lane in two instructions, the second fetch pays for the line
// again (6 lines per point and level with the pair gathers above). Here lanes 2i and 2i + 1 work on ONE point at a time — the
// even lane fetches the four lo-x corners, the odd lane the four hi-x corners, in the same four instructions — first on the even
// lane's point, then on the odd lane's: 8 gather instructions of 32 points x ~1.06 lines instead of 4 x 64 + 4 x 32, i.e.
// 4.25 lines per point and level. The neighbour's cell hashes and the fetched values cross the lane pair as DPP quad_perm
// moves (VALU rate, no LDS); every lane then blends its own point exactly as before: same operations, same bits.
// (Two points per lane on top of it — twice the gathers in flight, half the waves — changes nothing: 63.0 against 63.1 us,
// profiles/r05_s17_*. What is left of a coarse level, 1.5 us, is the ~300 vector instructions of a (point, level) thread.)
template <bool kXcd>
__global__ __launch_bounds__(kHashBlock) void hash_encode_fwd_v3_kernel(nsamd_points P, int64_t M, int transform,
                                                                        nsamd_aabb box, const float2* __restrict__ table,
                                                                        nsamd_grid grid, float* __restrict__ enc,
                                                                        int64_t stride_p, int64_t stride_k,
                                                                        float* __restrict__ selector, unsigned nb) {
  int level;
  int64_t pb;
  if (kXcd) {
    const unsigned xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const unsigned li = q / nb;
    pb = q - li * nb;
    level = (li & 1u) ? (int)(8u * li + 7u - xcd) : (int)(8u * li + xcd);
    if (level >= grid.num_levels) return;  // (block-uniform)
  } else {
    level = blockIdx.y;
    pb = blockIdx.x;
  }
  const int64_t p = pb * kHashBlock + threadIdx.x;
  const bool live = p < M;
  float x, y, z;
  load_position_burst(P, live ? p : M - 1, x, y, z);  // (every lane stays: its neighbour needs it for the exchange)
  const float sel = normalise_position(transform, box, x, y, z);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const Cell c = locate_cell(x, y, z, grid.scalings[level]);
  const float2* __restrict__ tl = table + ((size_t)level << grid.log2_table_size);
  const bool odd = (threadIdx.x & 1u) != 0u;
  // this lane's x index in either round, the y / z hash terms of the point the round works on
  const uint32_t hy0 = (uint32_t)c.lo[1] * kPrimeY, hy1 = (uint32_t)c.hi[1] * kPrimeY;
  const uint32_t hz0 = (uint32_t)c.lo[2] * kPrimeZ, hz1 = (uint32_t)c.hi[2] * kPrimeZ;
  // what the neighbour needs of this lane's cell: the x index of ITS side (even neighbour: lo, odd neighbour: hi) and the terms
  const uint32_t x_for_nb = odd ? (uint32_t)c.lo[0] : (uint32_t)c.hi[0];  // (an odd lane's neighbour is even -> takes lo)
  const uint32_t nx = pair_swap_u32(x_for_nb);
  const uint32_t ny0 = pair_swap_u32(hy0), ny1 = pair_swap_u32(hy1), nz0 = pair_swap_u32(hz0), nz1 = pair_swap_u32(hz1);
  const uint32_t own_x = odd ? (uint32_t)c.hi[0] : (uint32_t)c.lo[0];
  // round E: the even lane's point (own for even lanes, the neighbour's for odd ones); round O: the odd lane's point
  const uint32_t ex = odd ? nx : own_x, ey0 = odd ? ny0 : hy0, ey1 = odd ? ny1 : hy1, ez0 = odd ? nz0 : hz0, ez1 = odd ? nz1 : hz1;
  const uint32_t ox = odd ? own_x : nx, oy0 = odd ? hy0 : ny0, oy1 = odd ? hy1 : ny1, oz0 = odd ? hz0 : nz0, oz1 = odd ? hz1 : nz1;
  // corner pair q: (y, z) = (lo, lo), (hi, lo), (lo, hi), (hi, hi) — v0/v1, v2/v3, v4/v5, v6/v7 of the kernels above
  const float2 e0 = tl[(ex ^ ey0 ^ ez0) & mask], e1 = tl[(ex ^ ey1 ^ ez0) & mask], e2 = tl[(ex ^ ey0 ^ ez1) & mask],
               e3 = tl[(ex ^ ey1 ^ ez1) & mask];
  const float2 o0 = tl[(ox ^ oy0 ^ oz0) & mask], o1 = tl[(ox ^ oy1 ^ oz0) & mask], o2 = tl[(ox ^ oy0 ^ oz1) & mask],
               o3 = tl[(ox ^ oy1 ^ oz1) & mask];
  // hand the neighbour what was fetched for ITS point (even lanes: round O, odd lanes: round E), keep the own round
  auto swap2 = [&](const float2& a, const float2& b) {
    const float2 send = odd ? a : b;  // (a: round E value, b: round O value)
    return make_float2(pair_swap_f32(send.x), pair_swap_f32(send.y));
  };
  const float2 r0 = swap2(e0, o0), r1 = swap2(e1, o1), r2 = swap2(e2, o2), r3 = swap2(e3, o3);
  // own point's corners: even lane = (lo-x: own round E, hi-x: received), odd lane = (lo-x: received, hi-x: own round O)
  const float2 v0 = odd ? r0 : e0, v1 = odd ? o0 : r0;
  const float2 v2 = odd ? r1 : e1, v3 = odd ? o1 : r1;
  const float2 v4 = odd ? r2 : e2, v5 = odd ? o2 : r2;
  const float2 v6 = odd ? r3 : e3, v7 = odd ? o3 : r3;
  const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
  const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
  float r[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    auto g = [&](const float2& a) { return f == 0 ? a.x : a.y; };
    const float yc_zc = g(v7) * wx + g(v6) * ux;  // blend order x, y, z exactly as encodings.py:446-456
    const float yf_zc = g(v5) * wx + g(v4) * ux;
    const float yf_zf = g(v1) * wx + g(v0) * ux;
    const float yc_zf = g(v3) * wx + g(v2) * ux;
    const float zc = yc_zc * wy + yf_zc * uy;
    const float zf = yc_zf * wy + yf_zf * uy;
    r[f] = zc * wz + zf * uz;
  }
  if (!live) return;
  float* o = enc + p * stride_p + (int64_t)(2 * level) * stride_k;
  o[0] = r[0];
  o[stride_k] = r[1];
  if (level == 0 && selector != nullptr) selector[p] = sel;
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

// LDS accumulate of one (g0, g1) pair. Measured on MI355X (scripts/probe_lds_atomics.hip, profiles/): ds_add_f32 with
// divergent addresses retires only 0.33 lane-ops/clk/CU — 12x below ds_add_u32 / a plain LDS read-modify-write — but
// 2.4 when the lanes of a wave share an address; a compare-and-swap loop is the opposite (2.2 when divergent). So:
// ONE 64-bit CAS attempt on the float pair (random hashed addresses: almost always succeeds, and covers both features
// with a single LDS atomic), and only the lanes that lost a race — true same-entry conflicts, i.e. hot coarse cells —
// fall back to ds_add_f32, which is the fast path for exactly that case.
__device__ __forceinline__ void lds_add_pair(float* pair, float v0, float v1) {
  unsigned long long* w = reinterpret_cast<unsigned long long*>(pair);
  const unsigned long long old = *w;
  const float n0 = __uint_as_float((uint32_t)old) + v0;
  const float n1 = __uint_as_float((uint32_t)(old >> 32)) + v1;
  const unsigned long long want = (unsigned long long)__float_as_uint(n0) | ((unsigned long long)__float_as_uint(n1) << 32);
  if (atomicCAS(w, old, want) != old) {
    atomicAdd(pair + 0, v0);  // ds_add_f32
    atomicAdd(pair + 1, v1);
  }
}

// ---- partitioned scatter (see the header comment) ----------------------------------------------------------------
constexpr int kSliceLog2Max = 14;  // 16384 entries x 2 floats = 128 KiB of the 160 KiB LDS
constexpr int kSliceThreads = 1024;

__global__ __launch_bounds__(kSliceThreads) void hash_encode_bwd_sliced_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, float* __restrict__ dst, int64_t dst_chunk_stride, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [slice_entries][2]
  const int slice = blockIdx.x, level = blockIdx.y, chunk = blockIdx.z, chunks = gridDim.z;
  const int slice_log2 = min(grid.log2_table_size, kSliceLog2Max);
  const int slice_entries = 1 << slice_log2;
  for (int e = threadIdx.x; e < 2 * slice_entries; e += kSliceThreads) acc[e] = 0.0f;
  __syncthreads();
  const int64_t per = (M + chunks - 1) / chunks;
  const int64_t p_end = min(M, (int64_t)(chunk + 1) * per);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const float scale = grid.scalings[level];
  for (int64_t p = (int64_t)chunk * per + threadIdx.x; p < p_end; p += kSliceThreads) {
    const float* gptr = denc + p * stride_p + (int64_t)(2 * level) * stride_k;
    const float g0 = gptr[0], g1 = gptr[stride_k];
    if (g0 == 0.0f && g1 == 0.0f) continue;
    float x, y, z;
    load_position(P, p, x, y, z);
    (void)normalise_position(transform, box, x, y, z);
    const Cell c = locate_cell(x, y, z, scale);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t idx = corner_index(c, k, mask);
      if ((int)(idx >> slice_log2) == slice) {
        const float bz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
        const float by = (k & 2) ? c.w[1] : 1.0f - c.w[1];
        const float bx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
        const uint32_t local = idx & (uint32_t)(slice_entries - 1);
        lds_add_pair(acc + 2 * local, ((g0 * bz) * by) * bx, ((g1 * bz) * by) * bx);
      }
    }
  }
  __syncthreads();
  float* out = dst + (int64_t)chunk * dst_chunk_stride +
               ((((size_t)level << grid.log2_table_size) + ((size_t)slice << slice_log2)) << 1);
  if (accumulate) {
    for (int e = threadIdx.x; e < 2 * slice_entries; e += kSliceThreads) out[e] += acc[e];  // sole owner of the tile
  } else {
    for (int e = threadIdx.x; e < 2 * slice_entries; e += kSliceThreads) out[e] = acc[e];
  }
}

// dL/d(raw position) of one point: back through the trilinear blend (offset = scaled - floor(scaled), slope
// scalings[l]) of every level, the selector, the affine map and the contraction Jacobian — what autograd computes for
// the reference (SURVEY.md §8a gradient-flow facts). No atomics.
__device__ __forceinline__ void position_gradient(const nsamd_points& P, int64_t p, int transform, const nsamd_aabb& box,
                                                  const float2* __restrict__ table, const nsamd_grid& grid,
                                                  const float* __restrict__ denc, int64_t stride_p, int64_t stride_k,
                                                  float& gx, float& gy, float& gz) {
  float rx, ry, rz;
  load_position(P, p, rx, ry, rz);
  float x = rx, y = ry, z = rz;
  const float sel = normalise_position(transform, box, x, y, z);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  gx = gy = gz = 0.0f;
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
}

// dL/dposition per point [M,3]: one thread per point. Only needed when the camera optimiser or normals are on.
__global__ __launch_bounds__(kHashBlock) void hash_encode_bwd_pos_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, const float2* __restrict__ table, nsamd_grid grid,
    const float* __restrict__ denc, int64_t stride_p, int64_t stride_k, float* __restrict__ dpos) {
  const int64_t p = (int64_t)blockIdx.x * kHashBlock + threadIdx.x;
  if (p >= M) return;
  float gx, gy, gz;
  position_gradient(P, p, transform, box, table, grid, denc, stride_p, stride_k, gx, gy, gz);
  dpos[3 * p + 0] = gx;
  dpos[3 * p + 1] = gy;
  dpos[3 * p + 2] = gz;
}

// dL/d(origins, directions) per RAY (ray mode: position = o + d * (t_i + t_{i+1}) / 2, cameras/rays.py:50-59), what
// the camera optimiser needs (cameras/camera_optimizers.py:148-153 makes origins / directions functions of the pose
// correction): d_origin = sum_s dL/dp_s, d_direction = sum_s dL/dp_s * (t_s + t_{s+1}) / 2. One wavefront per ray, lane
// l takes samples l, l + 64, ... in order, the 64 partial sums meet in a fixed butterfly: bit-reproducible, and the
// [M,3] per-point gradient never touches HBM.
__global__ __launch_bounds__(256) void hash_encode_bwd_rays_kernel(
    nsamd_points P, int64_t num_rays, int transform, nsamd_aabb box, const float2* __restrict__ table, nsamd_grid grid,
    const float* __restrict__ denc, int64_t stride_p, int64_t stride_k, float* __restrict__ d_origins,
    float* __restrict__ d_directions, int accumulate, const uint32_t* __restrict__ gate,
    const uint8_t* __restrict__ ray_mask) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  if ((gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ||
      (ray_mask != nullptr && ray_mask[ray] == 0)) {
    // gated call, no gradient on this level: `denc` (not written) stands for zeros, so do the ray gradients
    if (!accumulate && lane < 3) {
      d_origins[3 * ray + lane] = 0.0f;
      d_directions[3 * ray + lane] = 0.0f;
    }
    return;
  }
  const int S = P.samples_per_ray;
  float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
  for (int smp = lane; smp < S; smp += 64) {
    const int64_t p = ray * S + smp;
    float gx, gy, gz;
    position_gradient(P, p, transform, box, table, grid, denc, stride_p, stride_k, gx, gy, gz);
    const float* tb = P.t_bins + ray * (S + 1) + smp;
    const float half = (tb[0] + tb[1]) / 2.0f;
    so[0] += gx; so[1] += gy; so[2] += gz;
    sd[0] += gx * half; sd[1] += gy * half; sd[2] += gz * half;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      so[k] += __shfl_xor(so[k], m);
      sd[k] += __shfl_xor(sd[k], m);
    }
  }
  if (lane < 3) {
    const float o = lane == 0 ? so[0] : (lane == 1 ? so[1] : so[2]);
    const float d = lane == 0 ? sd[0] : (lane == 1 ? sd[1] : sd[2]);
    if (accumulate) {
      d_origins[3 * ray + lane] += o;
      d_directions[3 * ray + lane] += d;
    } else {
      d_origins[3 * ray + lane] = o;
      d_directions[3 * ray + lane] = d;
    }
  }
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

// NeRFEncoding.pytorch_fwd without covariances (encodings.py:148-189): out[p] = [sin(s), sin(s + pi/2), x] with
// s[d * F + f] = (2 pi x_d) * freq_f, the raw input LAST (forward() :175-176). One thread per (point, d, f): both sines
// of its phase; the fp32 operation order is the reference's (scalar 2 pi rounded to fp32, product, sum, sin).
__global__ __launch_bounds__(256) void nerf_encode_kernel(nsamd_points P, int64_t M, const float* __restrict__ freqs, int F,
                                                         bool include_input, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per = 3 * F;
  if (e >= M * per) return;
  const int64_t p = e / per;
  const int k = (int)(e - p * per);
  const int d = k / F, f = k - d * F;
  float x[3];
  load_position(P, p, x[0], x[1], x[2]);
  const float xd = d == 0 ? x[0] : (d == 1 ? x[1] : x[2]);
  const float s = (6.283185307179586f * xd) * freqs[f];
  const int width = 2 * per + (include_input ? 3 : 0);
  float* o = out + p * width;
  o[k] = sinf(s);
  o[per + k] = sinf(s + 1.5707963267948966f);
  if (include_input && f == 0) o[2 * per + d] = xd;
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

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e != nullptr ? atoi(e) : dflt;
}

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
  // Levels per thread: small tables (all levels of a proposal grid sit in one XCD's L2 together) gain from sharing the
  // position and having 32 gathers in flight (44.5 -> 38.8 us, 27.9 -> 24.6 us); for the main grid one level already
  // fills an L2 and sweeping several at once thrashes it (80 -> 97 us). Measured, profiles/r01_negative_results.txt.
  static const int lv_force = env_int("NSAMD_HASH_FWD_LEVELS", 0);
  const int lv_env = lv_force ? lv_force : (((int64_t)8 << grid.log2_table_size) >= (2 << 20) ? 1 : 4);
  if (lv_env >= 4) {
    dim3 g((unsigned)nb, (unsigned)((grid.num_levels + 3) / 4));
    hash_encode_fwd_kernel<4><<<g, kHashBlock, 0, (hipStream_t)stream>>>(
        pts, M, transform, aabb, reinterpret_cast<const float2*>(table), grid, enc, stride_p, stride_k, selector);
  } else if (lv_env >= 2) {
    dim3 g((unsigned)nb, (unsigned)((grid.num_levels + 1) / 2));
    hash_encode_fwd_kernel<2><<<g, kHashBlock, 0, (hipStream_t)stream>>>(
        pts, M, transform, aabb, reinterpret_cast<const float2*>(table), grid, enc, stride_p, stride_k, selector);
  } else {
    // pair gathers + XCD-aware level sweep (mode 3): 83.8 -> 78.8 us on the main grid (81.5 with either alone); lane pairs on
    // one line + the XCD-aware sweep (mode 7, round 5): 76.1 -> 63.3 us on one box, a fine level 5.5 -> 4.4 us
    // (profiles/r05_s15_*). Same arithmetic in every mode.
    static const int mode = env_int("NSAMD_HASH_FWD_MODE", 7);
    const float2* t2 = reinterpret_cast<const float2*>(table);
    const dim3 g2((unsigned)nb, (unsigned)grid.num_levels);
    const int64_t xcd_blocks = 8 * nb * ((grid.num_levels + 7) / 8);
    const bool xcd = (mode & 2) && grid.num_levels % 8 == 0 && xcd_blocks <= 0x7fffffffLL;
    const dim3 g1((unsigned)xcd_blocks);
    if ((mode & 4) && xcd)
      hash_encode_fwd_v3_kernel<true><<<g1, kHashBlock, 0, (hipStream_t)stream>>>(
          pts, M, transform, aabb, t2, grid, enc, stride_p, stride_k, selector, (unsigned)nb);
    else if (mode & 4)
      hash_encode_fwd_v3_kernel<false><<<g2, kHashBlock, 0, (hipStream_t)stream>>>(
          pts, M, transform, aabb, t2, grid, enc, stride_p, stride_k, selector, (unsigned)nb);
    else if ((mode & 1) && xcd)
      hash_encode_fwd_v2_kernel<true, true><<<g1, kHashBlock, 0, (hipStream_t)stream>>>(
          pts, M, transform, aabb, t2, grid, enc, stride_p, stride_k, selector, (unsigned)nb);
    else if (mode & 1)
      hash_encode_fwd_v2_kernel<true, false><<<g2, kHashBlock, 0, (hipStream_t)stream>>>(
          pts, M, transform, aabb, t2, grid, enc, stride_p, stride_k, selector, (unsigned)nb);
    else if (xcd)
      hash_encode_fwd_v2_kernel<false, true><<<g1, kHashBlock, 0, (hipStream_t)stream>>>(
          pts, M, transform, aabb, t2, grid, enc, stride_p, stride_k, selector, (unsigned)nb);
    else
      hash_encode_fwd_kernel<1><<<g2, kHashBlock, 0, (hipStream_t)stream>>>(pts, M, transform, aabb, t2, grid, enc, stride_p,
                                                                          stride_k, selector);
  }
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

static int hashgrid_encode_bwd_impl(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                                    nsamd_grid grid, const float* denc, int64_t stride_p, int64_t stride_k,
                                    float* dtable, float* dpositions, float* workspace, int64_t workspace_floats,
                                    bool overwrite, const uint32_t* gate, const uint8_t* ray_mask, nsamd_stream_t stream) {
  if (M == 0 && !overwrite) return NSAMD_OK;
  int st = check_points(pts, M);
  if (st) return st;
  st = check_grid(grid);
  if (st) return st;
  NSAMD_REQUIRE(M == 0 || denc != nullptr);
  NSAMD_REQUIRE(transform >= 0 && transform <= 2);
  NSAMD_REQUIRE(dtable != nullptr || dpositions != nullptr);
  // The binned, order-independent scatter (scatter.hip) whenever the caller's scratch holds its plan
  ScatterPlan plan{};
  if (dtable != nullptr && M > 0 && workspace != nullptr) {
    plan = scatter_plan(grid, M, overwrite);
    if (plan.ok && plan.total_words > workspace_floats) plan.ok = false;
  }
  if (overwrite) {
    // write-only table gradient: the binned path overwrites every tile; anything else zero-fills first
    NSAMD_REQUIRE(dtable != nullptr);
    if (!plan.ok) {
      if (hipMemsetAsync(dtable, 0, sizeof(float) * 2 * ((size_t)grid.num_levels << grid.log2_table_size),
                         (hipStream_t)stream) != hipSuccess)
        return NSAMD_ERR_LAUNCH;
      overwrite = false;
    }
  }
  if (M == 0) return NSAMD_OK;
  const int64_t nb = (M + kHashBlock - 1) / kHashBlock;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  if (dtable != nullptr && plan.ok) {
    st = scatter_launch(pts, M, transform, aabb, grid, denc, stride_p, stride_k, dtable, workspace, plan, overwrite,
                        gate, ray_mask, (hipStream_t)stream);
    if (st) return st;
  } else if (gate != nullptr) {
    return NSAMD_ERR_INVALID_ARG;  // gated calls exist for the binned scatter only (the training step's workspaces)
  } else if (dtable != nullptr && M < 8192) {
    // scratch-free, small batch: direct fire-and-forget float atomics (sums in no fixed order)
    dim3 g((unsigned)nb, (unsigned)grid.num_levels);
    hash_encode_bwd_table_kernel<<<g, kHashBlock, 0, (hipStream_t)stream>>>(pts, M, transform, aabb, grid, denc,
                                                                             stride_p, stride_k, dtable);
    NSAMD_CHECK_LAUNCH();
  } else if (dtable != nullptr) {
    // scratch-free fallback: one workgroup OWNS a tile and scans all sample points (float sums in no fixed order)
    const int slice_log2 = grid.log2_table_size < kSliceLog2Max ? grid.log2_table_size : kSliceLog2Max;
    const int slices = 1 << (grid.log2_table_size - slice_log2);
    const size_t lds = sizeof(float) * 2 * ((size_t)1 << slice_log2);
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NSAMD_ERR_NO_DEVICE;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {  // the opt-in is per device
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_encode_bwd_sliced_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(float) << kSliceLog2Max) !=
          hipSuccess)
        return NSAMD_ERR_LAUNCH;
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    dim3 g((unsigned)slices, (unsigned)grid.num_levels, 1u);
    hash_encode_bwd_sliced_kernel<<<g, kSliceThreads, lds, (hipStream_t)stream>>>(
        pts, M, transform, aabb, grid, denc, stride_p, stride_k, dtable, 0, /*accumulate=*/1);
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

extern "C" int nsamd_hashgrid_encode_bwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                         const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                         int64_t stride_k, float* dtable, float* dpositions, float* workspace,
                                         int64_t workspace_floats, nsamd_stream_t stream) {
  return hashgrid_encode_bwd_impl(pts, M, transform, aabb, table, grid, denc, stride_p, stride_k, dtable, dpositions,
                                  workspace, workspace_floats, false, nullptr, nullptr, stream);
}

extern "C" int nsamd_hashgrid_encode_bwd_gated(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                               const float* table, nsamd_grid grid, const float* denc,
                                               int64_t stride_p, int64_t stride_k, float* dtable, float* workspace,
                                               int64_t workspace_floats, const uint32_t* gate,
                                               const uint8_t* ray_mask, nsamd_stream_t stream) {
  NSAMD_REQUIRE(gate != nullptr && dtable != nullptr && workspace != nullptr);
  return hashgrid_encode_bwd_impl(pts, M, transform, aabb, table, grid, denc, stride_p, stride_k, dtable, nullptr,
                                  workspace, workspace_floats, false, gate, ray_mask, stream);
}

extern "C" int nsamd_hashgrid_encode_bwd_set(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                             const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                             int64_t stride_k, float* dtable, float* dpositions, float* workspace,
                                             int64_t workspace_floats, nsamd_stream_t stream) {
  return hashgrid_encode_bwd_impl(pts, M, transform, aabb, table, grid, denc, stride_p, stride_k, dtable, dpositions,
                                  workspace, workspace_floats, true, nullptr, nullptr, stream);
}

extern "C" int nsamd_hashgrid_encode_bwd_rays(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                              const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                              int64_t stride_k, float* d_origins, float* d_directions, int accumulate,
                                              nsamd_stream_t stream) {
  return nsamd_hashgrid_encode_bwd_rays_gated(pts, M, transform, aabb, table, grid, denc, stride_p, stride_k, d_origins,
                                              d_directions, accumulate, nullptr, nullptr, stream);
}

extern "C" int nsamd_hashgrid_encode_bwd_rays_gated(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                                    const float* table, nsamd_grid grid, const float* denc,
                                                    int64_t stride_p, int64_t stride_k, float* d_origins,
                                                    float* d_directions, int accumulate, const uint32_t* gate,
                                                    const uint8_t* ray_mask, nsamd_stream_t stream) {
  if (M == 0) return NSAMD_OK;
  int st = check_points(pts, M);
  if (st) return st;
  st = check_grid(grid);
  if (st) return st;
  NSAMD_REQUIRE(pts.positions == nullptr);  // ray mode only: explicit positions have no origin / direction to credit
  NSAMD_REQUIRE(table != nullptr && denc != nullptr && d_origins != nullptr && d_directions != nullptr);
  NSAMD_REQUIRE(transform >= 0 && transform <= 2);
  const int64_t rays = M / pts.samples_per_ray;
  const int64_t nb = (rays + 3) / 4;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  hash_encode_bwd_rays_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(
      pts, rays, transform, aabb, reinterpret_cast<const float2*>(table), grid, denc, stride_p, stride_k, d_origins,
      d_directions, accumulate ? 1 : 0, gate, ray_mask);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int64_t nsamd_hashgrid_encode_bwd_workspace(nsamd_grid grid, int64_t M, int write_only) {
  if (M <= 0 || check_grid(grid) != NSAMD_OK) return 0;
  const ScatterPlan p = scatter_plan(grid, M, write_only != 0);
  return p.ok ? p.total_words : 0;
}

extern "C" int64_t nsamd_hashgrid_encode_bwd_workspace_state(nsamd_grid grid, int64_t M) {
  if (M <= 0 || check_grid(grid) != NSAMD_OK) return 0;
  const ScatterPlan p = scatter_plan(grid, M, false);
  return p.ok ? p.state_words : 0;
}

extern "C" int nsamd_hashgrid_scatter_events(const float* workspace, uint32_t* events_host, nsamd_stream_t stream) {
  NSAMD_REQUIRE(workspace != nullptr && events_host != nullptr);
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(workspace);
  if (hipMemcpyAsync(events_host, hdr + kHdrEvtSpill, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                     (hipStream_t)stream) != hipSuccess)
    return NSAMD_ERR_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return NSAMD_ERR_LAUNCH;
  return NSAMD_OK;
}

extern "C" int nsamd_sh4_encode(const float* dirs, int64_t M, float* out, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && (M == 0 || (dirs != nullptr && out != nullptr)));
  if (M == 0) return NSAMD_OK;
  sh4_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(dirs, M, out);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_nerf_encode(nsamd_points pts, int64_t M, const float* freqs, int32_t num_frequencies,
                                 int32_t include_input, float* out, nsamd_stream_t stream) {
  int st = check_points(pts, M);
  if (st) return st;
  NSAMD_REQUIRE(num_frequencies > 0 && num_frequencies <= 64);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(freqs != nullptr && out != nullptr);
  const int64_t threads = M * 3 * num_frequencies;
  const int64_t blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  nerf_encode_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(pts, M, freqs, num_frequencies, include_input != 0, out);
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
