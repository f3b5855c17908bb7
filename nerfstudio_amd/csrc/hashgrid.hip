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

// ---- binned scatter: pass 1 (route) + pass 2 (apply) -------------------------------------------------------------
constexpr int kBinThreads = 1024;
constexpr int kMaxBins = 4096;
struct LevelList {
  int8_t level[32];
  int count;
};

// Pass 1 for the COARSE levels (cell wider than the sample spacing, few distinct entries per workgroup).
// Consecutive samples of a ray share a cell there, all rays of a camera start in the same cells, and a level has few
// entries in total: emitted naively, pass 2 serialises thousands of LDS read-modify-writes on a handful of hot
// entries (measured: level 0 alone took as long as all 16 levels together). So:
//  * every thread walks kRunLen CONSECUTIVE samples and sums the 8 corner contributions in registers while the cell
//    stays the same (a run) — sequential, no cross-lane traffic (the earlier wave-level segmented scan of 16 values
//    cost ~1500 instructions per sample and made this kernel ALU-bound);
//  * a finished run is summed per table entry into a workgroup-wide LDS hash table (open addressing, bounded probing;
//    a full table sends the update out as a direct atomic), so the workgroup emits ONE single record per distinct
//    entry.
constexpr int kRunThreads = 256;
constexpr uint32_t kEmptyKey = 0xffffffffu;

// Where pass 1 cannot queue an update (queue or combining table full, pair straddling two tiles) the update must still
// arrive exactly. Accumulating call: a direct global atomic on the table gradient. Write-only call (`..._set`: pass 2
// OVERWRITES every tile, so nothing may be added before it): the update is appended to a deferred list sized for the
// worst case and applied by hash_bwd_deferred_kernel after pass 2.
constexpr int kDeferredLists = 64;  // sub-lists, each with its own counter: a single counter serialises at ~12 ns/atomic
struct Deferred {
  uint32_t* count;  // [kDeferredLists] (+ ticket word); nullptr: accumulate directly
  uint4* list;      // kDeferredLists x sub_cap records
  uint32_t sub_cap;
};

__device__ __forceinline__ void fallback_add(float* level_table, int level, uint32_t index, float v0, float v1,
                                             const Deferred& d) {
  if (d.count != nullptr) {
    // one returning atomic per wavefront and sub-list; a group that does not fit moves on to the next sub-list (the
    // lists together hold the worst case plus 64 records of slack each, so it always lands)
    const unsigned long long active = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __builtin_ctzll(active);
    const uint32_t n = (uint32_t)__builtin_popcountll(active);
    const uint32_t mine = (uint32_t)__builtin_popcountll(active & ((1ull << lane) - 1ull));
    for (int t = 0; t < kDeferredLists; ++t) {
      const uint32_t sub = (blockIdx.x + blockIdx.y * 7u + (uint32_t)t) & (kDeferredLists - 1);
      uint32_t base = 0u;
      if (lane == leader) base = atomicAdd(d.count + sub, n);
      base = __shfl(base, leader);
      if (base + n <= d.sub_cap) {
        d.list[(size_t)sub * d.sub_cap + base + mine] =
            make_uint4(index, (uint32_t)level, __float_as_uint(v0), __float_as_uint(v1));
        return;
      }
    }
  }
  unsafeAtomicAdd(level_table + 2 * (size_t)index, v0);
  unsafeAtomicAdd(level_table + 2 * (size_t)index + 1, v1);
}

// DPP moves inside a row of 16 lanes: value of lane - D (row_shr) / lane + 1 (row_shl); lanes without a source get `old`
template <int D>
__device__ __forceinline__ int dpp_row_shr(int v, int old) {
  return __builtin_amdgcn_update_dpp(old, v, 0x110 | D, 0xf, 0xf, false);
}
template <int D>
__device__ __forceinline__ float dpp_row_shr(float v, float old) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x110 | D, 0xf, 0xf, false));
}
__device__ __forceinline__ int dpp_row_shl1(int v, int old) {
  return __builtin_amdgcn_update_dpp(old, v, 0x101, 0xf, 0xf, false);
}

// One step of the segmented inclusive scan over the lanes of a 16-lane row (f = "my prefix already reaches a segment
// head").
template <int D>
__device__ __forceinline__ void lane_merge_step(float (&a0)[8], float (&a1)[8], int& f) {
  const int fp = dpp_row_shr<D>(f, 1);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t0 = dpp_row_shr<D>(a0[k], 0.0f), t1 = dpp_row_shr<D>(a1[k], 0.0f);
    if (!f) { a0[k] += t0; a1[k] += t1; }
  }
  if (!f) f = fp;
}

// kCombineBits: log2 of the table size. 12 (48 KiB, 3 workgroups per CU) holds the distinct entries of ~20 rays of a
// level with resolution < 64; 11 (24 KiB, 6 workgroups per CU) is enough when a workgroup's 1024 samples are only a
// few long rays (samples per ray >= 192) and doubles the occupancy of this latency-bound kernel.
template <int kRunLen, int kCombineBits>
__global__ __launch_bounds__(kRunThreads) void hash_bwd_bin_runs_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, int slice_log2, uint32_t cap, LevelList levels,
    uint32_t* __restrict__ cursors, uint4* __restrict__ queues, float* __restrict__ dtable, Deferred deferred) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  constexpr int kCombineSlots = 1 << kCombineBits;
  constexpr int kCombinePerThread = kCombineSlots / kRunThreads;
  const int level = levels.level[blockIdx.y];
  const int B = 1 << (grid.log2_table_size - slice_log2);
  // layout: [vals: 2 x slots floats][keys: slots][cnt: B][base: B]
  float* vals = reinterpret_cast<float*>(lds_u);
  uint32_t* keys = lds_u + 2 * kCombineSlots;
  uint32_t* cnt = lds_u + 3 * kCombineSlots;  // [B] records of this workgroup per tile
  uint32_t* base = cnt + B;                   // [B] reserved queue offset per tile
  for (int t = threadIdx.x; t < B; t += kRunThreads) cnt[t] = 0;
#pragma unroll
  for (int i = 0; i < kCombinePerThread; ++i) {
    const int sl = threadIdx.x + i * kRunThreads;
    keys[sl] = kEmptyKey;
    vals[2 * sl] = 0.0f;
    vals[2 * sl + 1] = 0.0f;
  }
  const int64_t p0 = ((int64_t)blockIdx.x * kRunThreads + threadIdx.x) * kRunLen;
  float g0[kRunLen], g1[kRunLen];
#pragma unroll
  for (int i = 0; i < kRunLen; ++i) {  // gradient loads in flight before anything depends on them
    g0[i] = 0.0f;
    g1[i] = 0.0f;
    if (p0 + i < M) {
      const float* gptr = denc + (p0 + i) * stride_p + (int64_t)(2 * level) * stride_k;
      g0[i] = gptr[0];
      g1[i] = gptr[stride_k];
    }
  }
  float px[kRunLen], py[kRunLen], pz[kRunLen];
#pragma unroll
  for (int i = 0; i < kRunLen; ++i) {  // ... and the positions: the run loop below must not wait on global memory
    px[i] = py[i] = pz[i] = 0.0f;
    if (p0 + i < M) {
      load_position(P, p0 + i, px[i], py[i], pz[i]);
      (void)normalise_position(transform, box, px[i], py[i], pz[i]);
    }
  }
  __syncthreads();
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  float* const level_table = dtable + (((size_t)level << grid.log2_table_size) << 1);
  const float scale = grid.scalings[level];
  Cell cur{};
  float a0[8], a1[8];
  bool have = false, single = true;  // single: no run of this thread has been flushed yet
#pragma unroll
  for (int i = 0; i <= kRunLen; ++i) {
    if (i == kRunLen) {
      // Lane-level run merging before the last flush: consecutive lanes are consecutive pieces of a ray, and on a
      // coarse level several of them sit in ONE cell. A lane whose only run continues the previous lane's open run
      // hands nothing to the table itself: the sums travel down the chain (segmented scan inside 16-lane rows, DPP
      // only) and the last lane of the chain inserts once. The table inserts are what bounds this kernel (LDS
      // address conflicts between exactly these lanes, profiles/r01_scatter_pmc_counters.log).
      int same = (have && single && (threadIdx.x & 15) != 0) ? 1 : 0;
      same &= dpp_row_shr<1>(have ? 1 : 0, 0);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        same &= (dpp_row_shr<1>(cur.lo[a], -1) == cur.lo[a]) ? 1 : 0;
        same &= (dpp_row_shr<1>(cur.hi[a], -1) == cur.hi[a]) ? 1 : 0;
      }
      const int head = same ^ 1;
      int f = head;
      lane_merge_step<1>(a0, a1, f);
      lane_merge_step<2>(a0, a1, f);
      lane_merge_step<4>(a0, a1, f);
      lane_merge_step<8>(a0, a1, f);
      have = have && (dpp_row_shl1(head, 1) != 0);  // only the last lane of a chain still owns a run
    }
    bool live = false;
    Cell c = cur;
    if (i < kRunLen) {
      live = (p0 + i < M) && !(g0[i] == 0.0f && g1[i] == 0.0f);
      if (live) c = locate_cell(px[i], py[i], pz[i], scale);
    }
    const bool same = have && live && c.lo[0] == cur.lo[0] && c.lo[1] == cur.lo[1] && c.lo[2] == cur.lo[2] &&
                      c.hi[0] == cur.hi[0] && c.hi[1] == cur.hi[1] && c.hi[2] == cur.hi[2];
    if (have && (i == kRunLen || (live && !same))) {  // the run ends: sum it into the workgroup's table
      // Each step below is one LDS round trip; the 8 corners go through every step together (8 operations in flight)
      // instead of one corner after the other — the workgroup's life is this latency chain.
      uint32_t index[8], h[8], prev[8];
      unsigned long long old[8];
      uint32_t placed = 0u;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        index[k] = corner_index(cur, k, mask);
        h[k] = (index[k] * 0x9E3779B1u) >> (32 - kCombineBits);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) prev[k] = atomicCAS(keys + h[k], kEmptyKey, index[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (prev[k] == kEmptyKey || prev[k] == index[k]) {
          placed |= 1u << k;
        } else {  // slot taken by another entry: linear probing (rare while the table is sparse)
          for (int probe = 1; probe < 8 && !((placed >> k) & 1u); ++probe) {
            h[k] = (h[k] + 1) & (kCombineSlots - 1);
            const uint32_t pv = atomicCAS(keys + h[k], kEmptyKey, index[k]);
            if (pv == kEmptyKey || pv == index[k]) placed |= 1u << k;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((placed >> k) & 1u) old[k] = *reinterpret_cast<volatile unsigned long long*>(vals + 2 * h[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if ((placed >> k) & 1u) {
          const float n0 = __uint_as_float((uint32_t)old[k]) + a0[k];
          const float n1 = __uint_as_float((uint32_t)(old[k] >> 32)) + a1[k];
          const unsigned long long want =
              (unsigned long long)__float_as_uint(n0) | ((unsigned long long)__float_as_uint(n1) << 32);
          old[k] = atomicCAS(reinterpret_cast<unsigned long long*>(vals + 2 * h[k]), old[k], want) ^ old[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if ((placed >> k) & 1u) {
          if (old[k] != 0ull) {  // lost the race on this slot (another run of the same cell): ds_add_f32
            atomicAdd(vals + 2 * h[k], a0[k]);
            atomicAdd(vals + 2 * h[k] + 1, a1[k]);
          }
        } else {  // table full around this slot: direct atomics keep the result exact
          fallback_add(level_table, level, index[k], a0[k], a1[k], deferred);
        }
      }
      have = false;
      single = false;
    }
    if (i < kRunLen && live) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float bz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
        const float by = (k & 2) ? c.w[1] : 1.0f - c.w[1];
        const float bx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
        const float t0 = ((g0[i] * bz) * by) * bx, t1 = ((g1[i] * bz) * by) * bx;
        a0[k] = same ? a0[k] + t0 : t0;
        a1[k] = same ? a1[k] + t1 : t1;
      }
      cur = c;
      have = true;
    }
  }
  __syncthreads();  // all sums of the workgroup are in the table
  uint32_t skey[kCombinePerThread], srank[kCombinePerThread];
#pragma unroll
  for (int i = 0; i < kCombinePerThread; ++i) {
    skey[i] = keys[threadIdx.x + i * kRunThreads];
    srank[i] = 0u;
    if (skey[i] != kEmptyKey) srank[i] = atomicAdd(cnt + (skey[i] >> slice_log2), 1u);  // ds_add_rtn_u32
  }
  __syncthreads();
  for (int t = threadIdx.x; t < B; t += kRunThreads) {
    const uint32_t n = cnt[t];
    base[t] = n ? atomicAdd(cursors + (size_t)level * B + t, n) : 0u;
  }
  __syncthreads();
  const uint32_t local_mask = (1u << slice_log2) - 1u;
#pragma unroll
  for (int i = 0; i < kCombinePerThread; ++i) {
    if (skey[i] == kEmptyKey) continue;
    const int sl = threadIdx.x + i * kRunThreads;
    const uint32_t bin = skey[i] >> slice_log2;
    const uint32_t pos = base[bin] + srank[i];
    const float v0 = vals[2 * sl], v1 = vals[2 * sl + 1];
    if (pos < cap) {  // single record (f0, f1, -, local index): one global_store_dwordx4
      queues[((size_t)level * B + bin) * cap + pos] =
          make_uint4(__float_as_uint(v0), __float_as_uint(v1), 0u, skey[i] & local_mask);
    } else {  // queue full (a very hot tile): direct atomics keep the result exact
      fallback_add(level_table, level, skey[i], v0, v1, deferred);
    }
  }
}

// Pass 1 for the FINE levels (no run-merging, no combining): every thread takes its point through kFineLevels (2 or 4)
// levels at once. A workgroup's life is a chain of latencies (gradient loads -> LDS ranks -> barrier -> one returning global
// atomic per tile -> barrier -> stores), and with 2048 threads per CU there is no occupancy left to hide it: levels
// per thread are the only source of independent work. The position (ray fetch + contraction) is also computed once
// instead of once per level. Emits x-pair records only.
template <int kFineLevels>
__global__ __launch_bounds__(kBinThreads) void hash_bwd_bin_fine_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, int slice_log2, uint32_t cap, LevelList levels,
    uint32_t* __restrict__ cursors, uint4* __restrict__ queues, float* __restrict__ dtable, Deferred deferred) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  const int B = 1 << (grid.log2_table_size - slice_log2);
  uint32_t* cnt = lds_u;                     // [kFineLevels][B]
  uint32_t* base = lds_u + kFineLevels * B;  // [kFineLevels][B]
  for (int t = threadIdx.x; t < kFineLevels * B; t += kBinThreads) cnt[t] = 0;
  const int first = blockIdx.y * kFineLevels;
  const int64_t p = (int64_t)blockIdx.x * kBinThreads + threadIdx.x;
  const bool inside = p < M;
  float g0[kFineLevels], g1[kFineLevels];
  int lvl[kFineLevels];
#pragma unroll
  for (int i = 0; i < kFineLevels; ++i) {  // all gradient loads in flight before anything depends on them
    lvl[i] = first + i < levels.count ? (int)levels.level[first + i] : -1;
    g0[i] = 0.0f;
    g1[i] = 0.0f;
    if (inside && lvl[i] >= 0) {
      const float* gptr = denc + p * stride_p + (int64_t)(2 * lvl[i]) * stride_k;
      g0[i] = gptr[0];
      g1[i] = gptr[stride_k];
    }
  }
  float x = 0.f, y = 0.f, z = 0.f;
  if (inside) {
    load_position(P, p, x, y, z);
    (void)normalise_position(transform, box, x, y, z);
  }
  __syncthreads();
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const uint32_t local_mask = (1u << slice_log2) - 1u;
  float w[kFineLevels][3];
  uint32_t word[kFineLevels][4], bin[kFineLevels][4], rank[kFineLevels][4];
  uint32_t recmask = 0u;  // bit 4 i + q
#pragma unroll
  for (int i = 0; i < kFineLevels; ++i) {
    if (lvl[i] < 0 || !inside || (g0[i] == 0.0f && g1[i] == 0.0f)) continue;
    const Cell c = locate_cell(x, y, z, grid.scalings[lvl[i]]);
    w[i][0] = c.w[0]; w[i][1] = c.w[1]; w[i][2] = c.w[2];
    float* const level_table = dtable + (((size_t)lvl[i] << grid.log2_table_size) << 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t ia = corner_index(c, 2 * q, mask), ib = corner_index(c, 2 * q + 1, mask);
      bin[i][q] = ia >> slice_log2;
      word[i][q] = (ia & local_mask) | ((ib & local_mask) << 14) | 0x80000000u;
      if ((ib >> slice_log2) == bin[i][q]) {
        recmask |= 1u << (4 * i + q);
        rank[i][q] = atomicAdd(cnt + i * B + bin[i][q], 1u);  // ds_add_rtn_u32
      } else {  // the pair straddles two tiles (needs a carry past bit slice_log2): rare, direct atomics
        const float bz = (q & 2) ? c.w[2] : 1.0f - c.w[2];
        const float by = (q & 1) ? c.w[1] : 1.0f - c.w[1];
        const float a0 = (g0[i] * bz) * by, a1 = (g1[i] * bz) * by;
        fallback_add(level_table, lvl[i], ia, a0 * (1.0f - c.w[0]), a1 * (1.0f - c.w[0]), deferred);
        fallback_add(level_table, lvl[i], ib, a0 * c.w[0], a1 * c.w[0], deferred);
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kFineLevels * B; t += kBinThreads) {
    const int i = t / B;
    const int level = first + i < levels.count ? (int)levels.level[first + i] : -1;
    const uint32_t n = cnt[t];
    base[t] = (n && level >= 0) ? atomicAdd(cursors + (size_t)level * B + (t - i * B), n) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kFineLevels; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!((recmask >> (4 * i + q)) & 1u)) continue;
      const float bz = (q & 2) ? w[i][2] : 1.0f - w[i][2];
      const float by = (q & 1) ? w[i][1] : 1.0f - w[i][1];
      const float a0 = (g0[i] * bz) * by, a1 = (g1[i] * bz) * by;
      const uint32_t pos = base[i * B + bin[i][q]] + rank[i][q];
      if (pos < cap) {  // one 16-B record = one global_store_dwordx4
        queues[((size_t)lvl[i] * B + bin[i][q]) * cap + pos] =
            make_uint4(__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(w[i][0]), word[i][q]);
      } else {  // queue full (a very hot cell): direct atomics keep the result exact
        float* const level_table = dtable + (((size_t)lvl[i] << grid.log2_table_size) << 1);
        const size_t ia = ((size_t)bin[i][q] << slice_log2) + (word[i][q] & 0x3fffu);
        const size_t ib = ((size_t)bin[i][q] << slice_log2) + ((word[i][q] >> 14) & 0x3fffu);
        fallback_add(level_table, lvl[i], (uint32_t)ia, a0 * (1.0f - w[i][0]), a1 * (1.0f - w[i][0]), deferred);
        fallback_add(level_table, lvl[i], (uint32_t)ib, a0 * w[i][0], a1 * w[i][0], deferred);
      }
    }
  }
}

// Pass 2: one workgroup per (level, tile) streams the tile's queue into LDS and adds the finished tile to the table.
__global__ void hash_bwd_apply_kernel(nsamd_grid grid, int slice_log2, uint32_t cap, int level0,
                                      const uint32_t* __restrict__ cursors, const uint4* __restrict__ queues,
                                      float* __restrict__ dtable, int overwrite) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int bin = blockIdx.x, level = level0 + blockIdx.y;
  const int B = gridDim.x;
  const int entries = 1 << slice_log2;
  for (int e = threadIdx.x; e < 2 * entries; e += blockDim.x) acc[e] = 0.0f;
  __syncthreads();
  const uint32_t n = min(cursors[(size_t)level * B + bin], cap);
  __syncthreads();
  // self-cleaning cursor: the next launch finds zeros again, so no memset node is needed per call (the workspace is
  // zero-initialised once by its owner)
  if (threadIdx.x == 0) const_cast<uint32_t*>(cursors)[(size_t)level * B + bin] = 0u;
  const uint4* q = queues + ((size_t)level * B + bin) * cap;
  // The tile fills the LDS (one or two workgroups per CU), so memory-level parallelism has to come from each thread:
  // 8 records in flight before touching the LDS. A thread takes 2 groups of 4 CONSECUTIVE records (one 64-B line
  // each): consecutive samples of a ray sit next to each other in the queue and — where the sampler has concentrated
  // them — hit the same entries, so equal neighbours are summed in registers first (fewer LDS operations, and the
  // lanes of a wave no longer race each other on them; a lost race costs the slow divergent ds_add_f32 path).
  auto add_entry = [&](uint32_t word, float c0, float c1, float c2, float c3) {
    if (word & 0x80000000u) {  // x-pair
      lds_add_pair(acc + 2 * (word & 0x3fffu), c0, c1);
      lds_add_pair(acc + 2 * ((word >> 14) & 0x3fffu), c2, c3);
    } else {
      lds_add_pair(acc + 2 * word, c0, c1);
    }
  };
  const uint32_t per_pass = blockDim.x * 8u;
  for (uint32_t e0 = 0; e0 < n; e0 += per_pass) {
    uint4 r[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t e = e0 + (uint32_t)(u >> 2) * (blockDim.x * 4u) + threadIdx.x * 4u + (uint32_t)(u & 3);
      ok[u] = e < n;
      r[u] = ok[u] ? q[e] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      uint32_t word = 0u;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
      bool have = false;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int u = 4 * grp + v;
        if (!ok[u]) continue;
        const float f0 = __uint_as_float(r[u].x), f1 = __uint_as_float(r[u].y);
        float p0 = f0, p1 = f1, p2 = 0.f, p3 = 0.f;
        if (r[u].w & 0x80000000u) {  // x-pair: see pass 1
          const float wx = __uint_as_float(r[u].z), omx = 1.0f - wx;
          p0 = f0 * omx; p1 = f1 * omx; p2 = f0 * wx; p3 = f1 * wx;
        }
        if (have && r[u].w == word) {
          c0 += p0; c1 += p1; c2 += p2; c3 += p3;
        } else {
          if (have) add_entry(word, c0, c1, c2, c3);
          word = r[u].w; c0 = p0; c1 = p1; c2 = p2; c3 = p3;
          have = true;
        }
      }
      if (have) add_entry(word, c0, c1, c2, c3);
    }
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(
      dtable + ((((size_t)level << grid.log2_table_size) + ((size_t)bin << slice_log2)) << 1));
  const float4* a4 = reinterpret_cast<const float4*>(acc);
  if (overwrite) {  // write-only gradient: no zero-fill before the call, no read here
    for (int i = threadIdx.x; i < entries / 2; i += blockDim.x) out[i] = a4[i];
    return;
  }
  for (int i = threadIdx.x; i < entries / 2; i += blockDim.x) {  // sole owner of the tile: plain read-modify-write
    float4 o = out[i];
    const float4 a = a4[i];
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    out[i] = o;
  }
}

// Applies the deferred updates of a write-only call after pass 2 (normally none: one workgroup, returns at once) and
// leaves the counter at zero for the next call.
__global__ void hash_bwd_deferred_kernel(nsamd_grid grid, uint32_t* __restrict__ count, const uint4* __restrict__ list,
                                         uint32_t sub_cap, float* __restrict__ dtable) {
  for (int sub = 0; sub < kDeferredLists; ++sub) {
    // a counter can overshoot by groups that moved on to the next list: entries [0, first overshooting base) are valid,
    // and every group checked base + n <= sub_cap before writing, so clamping is exact for the written prefix only if
    // groups are written in counter order — they are not; instead every slot a group skipped stays "empty" (level
    // word = 0xffffffff, set by the previous pass of this kernel)
    const uint32_t n = min(count[sub], sub_cap);
    const uint4* l = list + (size_t)sub * sub_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const uint4 r = l[i];
      if (r.y == 0xffffffffu) continue;
      float* t = dtable + ((((size_t)r.y << grid.log2_table_size) + r.x) << 1);
      unsafeAtomicAdd(t, __uint_as_float(r.z));
      unsafeAtomicAdd(t + 1, __uint_as_float(r.w));
      const_cast<uint4*>(l)[i].y = 0xffffffffu;
    }
  }
  __syncthreads();
  // count[kDeferredLists] = ticket: the last workgroup to finish (every workgroup has read the counters by then) resets
  if (threadIdx.x == 0 && atomicAdd(count + kDeferredLists, 1u) == gridDim.x - 1) {
    for (int sub = 0; sub <= kDeferredLists; ++sub) count[sub] = 0u;
  }
}

// dtable[i] += sum over chunks of partial[c][i]
__global__ void hash_partial_reduce_kernel(const float* __restrict__ partial, int chunks, int64_t n,
                                           float* __restrict__ dtable) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(dtable)[i];
    for (int c = 0; c < chunks; ++c) {
      const float4 v = reinterpret_cast<const float4*>(partial + (int64_t)c * n)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dtable)[i] = s;
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
    dim3 g((unsigned)nb, (unsigned)grid.num_levels);
    hash_encode_fwd_kernel<1><<<g, kHashBlock, 0, (hipStream_t)stream>>>(
        pts, M, transform, aabb, reinterpret_cast<const float2*>(table), grid, enc, stride_p, stride_k, selector);
  }
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

// tiles (= bins x levels) the binned scatter aims for; tunable for experiments through NSAMD_SCATTER_TILES
static int scatter_target_tiles() {
  static int cached = 0;
  if (cached == 0) {
    const char* e = getenv("NSAMD_SCATTER_TILES");
    cached = (e != nullptr && atoi(e) >= 64) ? atoi(e) : 512;
  }
  return cached;
}


// Geometry of the binned scatter for (grid, M): tile size chosen so that (tiles = bins x levels) >= ~512 fills the
// chip; queues sized for 2x the uniform-hash expectation of 8 M single records per level (the fine levels need half:
// x-pair records), 4 words per record.
struct ScatterPlan {
  bool ok;
  int slice_log2, bins;
  int64_t tiles, cursor_words;
  uint32_t cap;
  int64_t deferred_cap;  // records of the deferred list (write-only calls), 0 = none
};

static bool sl_too_wide(int sl) { return sl > 14; }  // local indices are 14-bit

static ScatterPlan scatter_geometry(const nsamd_grid& grid) {
  ScatterPlan p{};
  int bits = 0;
  while ((grid.num_levels << bits) < scatter_target_tiles()) ++bits;
  int sl = grid.log2_table_size - bits;
  sl = sl > kSliceLog2Max ? kSliceLog2Max : (sl < 8 ? 8 : sl);
  if (sl > grid.log2_table_size) sl = grid.log2_table_size;
  p.slice_log2 = sl;
  p.bins = 1 << (grid.log2_table_size - sl);
  p.tiles = (int64_t)p.bins * grid.num_levels;
  p.cursor_words = (p.tiles + 3) & ~(int64_t)3;
  return p;
}

static int64_t scatter_expected_records(const ScatterPlan& p, int64_t M) { return (8 * M + p.bins - 1) / p.bins; }

// worst case of deferred updates: every corner update of the call; split over kDeferredLists sub-lists (+ 64 records of
// slack each: a wavefront's group needs contiguous room)
static int64_t scatter_deferred_cap(const nsamd_grid& grid, int64_t M) { return 8 * M * grid.num_levels; }
static int64_t scatter_deferred_sub_cap(int64_t total) { return (total + kDeferredLists - 1) / kDeferredLists + 64; }
constexpr int kDeferredCountWords = kDeferredLists + 4;  // counters + ticket, padded to 16 B

// workspace = [cursors: cursor_words][deferred counters: 68 words][queues: tiles x cap x 4]
//             [deferred lists: kDeferredLists x sub_cap x 4]
static ScatterPlan scatter_plan(const nsamd_grid& grid, int64_t M, const float* workspace, int64_t workspace_floats,
                                bool overwrite) {
  ScatterPlan p = scatter_geometry(grid);
  if (workspace == nullptr || p.bins > kMaxBins || sl_too_wide(p.slice_log2)) return p;
  p.deferred_cap = overwrite ? scatter_deferred_cap(grid, M) : 0;
  if (scatter_deferred_sub_cap(p.deferred_cap) > 0x7fffffffLL) return p;
  const int64_t deferred_words = overwrite ? 4 * kDeferredLists * scatter_deferred_sub_cap(p.deferred_cap) : 0;
  const int64_t cap = (workspace_floats - p.cursor_words - kDeferredCountWords - deferred_words) / (4 * p.tiles);
  const int64_t expect = scatter_expected_records(p, M);
  p.ok = cap >= expect + expect / 4 && cap < 0x7fffffffLL;
  p.cap = p.ok ? (uint32_t)cap : 0u;
  return p;
}

static int device_cus() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cached;
}

static int hashgrid_encode_bwd_impl(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                                    nsamd_grid grid, const float* denc, int64_t stride_p, int64_t stride_k,
                                    float* dtable, float* dpositions, float* workspace, int64_t workspace_floats,
                                    bool overwrite, nsamd_stream_t stream) {
  if (M == 0 && !overwrite) return NSAMD_OK;
  int st = check_points(pts, M);
  if (st) return st;
  st = check_grid(grid);
  if (st) return st;
  NSAMD_REQUIRE(denc != nullptr);
  NSAMD_REQUIRE(transform >= 0 && transform <= 2);
  NSAMD_REQUIRE(dtable != nullptr || dpositions != nullptr);
  if (overwrite) {
    // write-only table gradient: the binned path overwrites every tile; anything else zero-fills first
    NSAMD_REQUIRE(dtable != nullptr);
    if (M < 8192 || !scatter_plan(grid, M, workspace, workspace_floats, true).ok) {
      if (hipMemsetAsync(dtable, 0, sizeof(float) * 2 * ((size_t)grid.num_levels << grid.log2_table_size),
                         (hipStream_t)stream) != hipSuccess)
        return NSAMD_ERR_LAUNCH;
      overwrite = false;
    }
  }
  if (M == 0) return NSAMD_OK;
  const int64_t nb = (M + kHashBlock - 1) / kHashBlock;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  if (dtable != nullptr && M < 8192) {
    // small batches: direct fire-and-forget atomics (zeroing / writing whole tiles would dominate)
    dim3 g((unsigned)nb, (unsigned)grid.num_levels);
    hash_encode_bwd_table_kernel<<<g, kHashBlock, 0, (hipStream_t)stream>>>(pts, M, transform, aabb, grid, denc,
                                                                             stride_p, stride_k, dtable);
    NSAMD_CHECK_LAUNCH();
  } else if (dtable != nullptr && scatter_plan(grid, M, workspace, workspace_floats, overwrite).ok) {
    const ScatterPlan plan = scatter_plan(grid, M, workspace, workspace_floats, overwrite);
    const int sl = plan.slice_log2, B = plan.bins;
    const uint32_t cap = plan.cap;
    uint32_t* cursors = reinterpret_cast<uint32_t*>(workspace);
    uint4* queues = reinterpret_cast<uint4*>(cursors + plan.cursor_words + kDeferredCountWords);  // 16-B aligned
    Deferred deferred{nullptr, nullptr, 0u};
    if (overwrite) {
      deferred.count = cursors + plan.cursor_words;
      deferred.list = queues + (size_t)plan.tiles * cap;
      deferred.sub_cap = (uint32_t)scatter_deferred_sub_cap(plan.deferred_cap);
    }
    hipStream_t st = (hipStream_t)stream;
    // Coarse levels go through the run-merging / combining kernel: those whose cells are wide against the sample
    // spacing. Measured on MI355X (bench workload, profiles/r01_scatter_*): resolution < samples per ray / 2 (at least
    // 24) — 16, 22 of the main grid (S = 48), 16, 32 of the second proposal grid (S = 96); finer levels overflow the
    // workgroup's 4096-slot table and every overflow is a deferred / direct atomic. With >= 192 samples per ray a
    // workgroup holds only 4 rays and every level of the (small) proposal grid pays off.
    static const int combine_env = env_int("NSAMD_SCATTER_COMBINE_RES", 0);  // > 0 overrides the rule (experiments)
    float coarse_below = 24.0f;
    if (pts.positions == nullptr) {
      const float S = (float)pts.samples_per_ray;
      coarse_below = fmaxf(24.0f, 0.5f * S);
      if (pts.samples_per_ray >= 192) coarse_below = S;
    }
    if (combine_env > 0) coarse_below = (float)combine_env;
    uint32_t coarse_mask = 0;
    for (int l = 0; l < grid.num_levels; ++l)
      if (grid.scalings[l] < coarse_below) coarse_mask |= 1u << l;
    static const int only_env = env_int("NSAMD_SCATTER_ONLY_LEVEL", -1);  // diagnostics: a single level
    const int l0 = only_env >= 0 && only_env < grid.num_levels ? only_env : 0;
    const int nl = only_env >= 0 && only_env < grid.num_levels ? 1 : grid.num_levels;
    LevelList coarse{}, fine{};
    for (int l = l0; l < l0 + nl; ++l) {
      LevelList& dst = ((coarse_mask >> l) & 1u) ? coarse : fine;
      dst.level[dst.count++] = (int8_t)l;
    }
    static bool attr2 = false;
    if (!attr2) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_bwd_apply_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(float) << kSliceLog2Max) !=
          hipSuccess)
        return NSAMD_ERR_LAUNCH;
      attr2 = true;
    }
    const unsigned threads = sl > 11 ? 1024u : 256u;
    const unsigned point_blocks = (unsigned)((M + kBinThreads - 1) / kBinThreads);
    static const int fine_env = env_int("NSAMD_SCATTER_FINE_LEVELS", 4);
    if (fine.count > 0 && fine_env >= 4) {
      dim3 g1(point_blocks, (unsigned)((fine.count + 3) / 4));
      hash_bwd_bin_fine_kernel<4><<<g1, kBinThreads, sizeof(uint32_t) * 2 * 4 * (size_t)B, st>>>(
          pts, M, transform, aabb, grid, denc, stride_p, stride_k, sl, cap, fine, cursors, queues, dtable, deferred);
      NSAMD_CHECK_LAUNCH();
    } else if (fine.count > 0 && fine_env >= 2) {
      dim3 g1(point_blocks, (unsigned)((fine.count + 1) / 2));
      hash_bwd_bin_fine_kernel<2><<<g1, kBinThreads, sizeof(uint32_t) * 2 * 2 * (size_t)B, st>>>(
          pts, M, transform, aabb, grid, denc, stride_p, stride_k, sl, cap, fine, cursors, queues, dtable, deferred);
      NSAMD_CHECK_LAUNCH();
    } else if (fine.count > 0) {
      dim3 g1(point_blocks, (unsigned)fine.count);
      hash_bwd_bin_fine_kernel<1><<<g1, kBinThreads, sizeof(uint32_t) * 2 * (size_t)B, st>>>(
          pts, M, transform, aabb, grid, denc, stride_p, stride_k, sl, cap, fine, cursors, queues, dtable, deferred);
      NSAMD_CHECK_LAUNCH();
    }
    if (coarse.count > 0) {
      static const int bits_env = env_int("NSAMD_SCATTER_TABLE_BITS", 0);  // experiments: force 11 / 12
      // (an 11-bit table doubles the occupancy but overflows on dense gradients: 264 vs 275 us when it fits, 1153 us
      // when it does not, profiles/r01_scatter_runs_kernel_sweeps.log)
      const int bits = bits_env ? bits_env : 12;
      const size_t bin_lds = sizeof(uint32_t) * (2 * (size_t)B + 3 * ((size_t)1 << bits));
      const int64_t per_block = (int64_t)kRunThreads * 4;
      dim3 g1((unsigned)((M + per_block - 1) / per_block), (unsigned)coarse.count);
      if (bits == 11)
        hash_bwd_bin_runs_kernel<4, 11><<<g1, kRunThreads, bin_lds, st>>>(pts, M, transform, aabb, grid, denc, stride_p,
                                                                          stride_k, sl, cap, coarse, cursors, queues, dtable, deferred);
      else
        hash_bwd_bin_runs_kernel<4, 12><<<g1, kRunThreads, bin_lds, st>>>(pts, M, transform, aabb, grid, denc, stride_p,
                                                                          stride_k, sl, cap, coarse, cursors, queues, dtable, deferred);
      NSAMD_CHECK_LAUNCH();
    }
    dim3 g2((unsigned)B, (unsigned)nl);
    hash_bwd_apply_kernel<<<g2, threads, sizeof(float) * 2 * ((size_t)1 << sl), st>>>(grid, sl, cap, l0, cursors,
                                                                                    queues, dtable, overwrite ? 1 : 0);
    NSAMD_CHECK_LAUNCH();
    if (overwrite) {
      hash_bwd_deferred_kernel<<<256, 256, 0, st>>>(grid, deferred.count, deferred.list, deferred.sub_cap, dtable);
      NSAMD_CHECK_LAUNCH();
    }
  } else if (dtable != nullptr) {
    const int slice_log2 = grid.log2_table_size < kSliceLog2Max ? grid.log2_table_size : kSliceLog2Max;
    const int slices = 1 << (grid.log2_table_size - slice_log2);
    const size_t lds = sizeof(float) * 2 * ((size_t)1 << slice_log2);
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_encode_bwd_sliced_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(float) << kSliceLog2Max) !=
          hipSuccess)
        return NSAMD_ERR_LAUNCH;
      attr_set = true;
    }
    const int64_t table_floats = ((int64_t)grid.num_levels << grid.log2_table_size) * 2;
    // split the points into chunks until the grid covers ~2 workgroups per CU (needs workspace for the partials)
    int chunks = 1;
    const int tiles = slices * grid.num_levels;
    if (workspace != nullptr && tiles < 2 * device_cus()) {
      chunks = (2 * device_cus() + tiles - 1) / tiles;
      const int64_t fit = workspace_floats / table_floats;
      if (chunks > fit) chunks = (int)fit;
      if (chunks > 32) chunks = 32;
      const int64_t max_by_points = (M + 4095) / 4096;  // keep >= 4096 points per chunk
      if (chunks > max_by_points) chunks = (int)max_by_points;
      if (chunks < 1) chunks = 1;
    }
    dim3 g((unsigned)slices, (unsigned)grid.num_levels, (unsigned)chunks);
    if (chunks == 1) {
      hash_encode_bwd_sliced_kernel<<<g, kSliceThreads, lds, (hipStream_t)stream>>>(
          pts, M, transform, aabb, grid, denc, stride_p, stride_k, dtable, 0, /*accumulate=*/1);
      NSAMD_CHECK_LAUNCH();
    } else {
      hash_encode_bwd_sliced_kernel<<<g, kSliceThreads, lds, (hipStream_t)stream>>>(
          pts, M, transform, aabb, grid, denc, stride_p, stride_k, workspace, table_floats, /*accumulate=*/0);
      NSAMD_CHECK_LAUNCH();
      const unsigned rb = (unsigned)((table_floats / 4 + 255) / 256 < 4096 ? (table_floats / 4 + 255) / 256 : 4096);
      hash_partial_reduce_kernel<<<rb, 256, 0, (hipStream_t)stream>>>(workspace, chunks, table_floats, dtable);
      NSAMD_CHECK_LAUNCH();
    }
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
                                  workspace, workspace_floats, false, stream);
}

extern "C" int nsamd_hashgrid_encode_bwd_set(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                             const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                             int64_t stride_k, float* dtable, float* dpositions, float* workspace,
                                             int64_t workspace_floats, nsamd_stream_t stream) {
  return hashgrid_encode_bwd_impl(pts, M, transform, aabb, table, grid, denc, stride_p, stride_k, dtable, dpositions,
                                  workspace, workspace_floats, true, stream);
}

extern "C" int64_t nsamd_hashgrid_encode_bwd_workspace(nsamd_grid grid, int64_t M, int write_only) {
  if (M < 8192 || check_grid(grid) != NSAMD_OK) return 0;
  const ScatterPlan p = scatter_geometry(grid);
  if (p.bins > kMaxBins) return 0;
  const int64_t cap = 2 * scatter_expected_records(p, M) + 64;
  return p.cursor_words + kDeferredCountWords + 4 * p.tiles * cap +
         (write_only ? 4 * kDeferredLists * scatter_deferred_sub_cap(scatter_deferred_cap(grid, M)) : 0);
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
