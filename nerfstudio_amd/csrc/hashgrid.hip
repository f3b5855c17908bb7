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
// HBM-bound integer/gather work, no MFMA. Algorithmic bytes: 8 corners x 8 B per point and level (fwd),
// 8 corners x 16 B read-modify-write (bwd).
//
// Backward = scatter-add into the table gradient. Measured on MI355X (scripts/probe_scatter.py, profiles/): fp32
// global atomics retire at a flat ~20 G lane-ops/s chip-wide whatever the locality (they are serviced memory-side,
// past the per-XCD L2s; contended coarse levels drop to 6 G/s) — 15x below the gather rate, 2.9 ms for the nerfacto
// main table. So the big scatter uses NO global atomics: the table gradient is partitioned into (level, 16K-entry
// slice) tiles of 128 KiB; one 1024-thread workgroup OWNS a tile in LDS, scans the sample points, adds the corner
// contributions that hash into its slice with LDS atomics (ds_add_f32: thousands of lane-ops per clock chip-wide),
// and writes the tile back with plain coalesced stores. Re-deriving the 8 hashes per (point, level) once per slice
// costs ALU (measured 0.70 ms for the main table, 32 slices per level). With scratch memory from the caller the
// redundancy goes away too ("binned" path, the default): pass 1 derives every corner update ONCE and appends
// a 16-B record (local index, g0, g1, pad) to the queue of the tile it falls into — a workgroup-local counting sort in LDS, one
// returning global atomic per (workgroup, non-empty tile) to reserve queue space, one dwordx4 store per record; pass
// 2 runs one workgroup per tile that streams its queue into the LDS tile and stores it. Queues are sized 2x the
// uniform-hash expectation; the rare overflow falls back to a direct atomic, so the result never depends on sizing.
// Tiny problems keep the direct-atomic kernel.
#include <stdlib.h>

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

// Pass 1. `merge_mask` bit l set = on level l consecutive lanes (consecutive samples of a ray) are likely to share a
// cell (cell size > sample spacing): their 8-corner contributions are summed with a wave-level segmented scan over
// runs of identical cells and only the last lane of each run emits records. `combine_mask` bit l set = the surviving
// updates of the workgroup are additionally summed per table entry in a small LDS hash table (open addressing, bounded
// probing, overflow goes out as plain records), so the workgroup emits ONE record per distinct entry. Coarse levels
// have few entries and every ray of a camera starts in the same cells: without this, pass 2 serialises thousands of
// LDS read-modify-writes on a handful of hot entries (measured: level 0 alone took as long as all 16 levels).
constexpr int kCombineBits = 12;
constexpr int kCombineSlots = 1 << kCombineBits;
constexpr int kCombinePerThread = kCombineSlots / kBinThreads;
constexpr uint32_t kEmptyKey = 0xffffffffu;

__global__ __launch_bounds__(kBinThreads) void hash_bwd_bin_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, int slice_log2, uint32_t cap, uint32_t merge_mask, uint32_t combine_mask,
    int level0, uint32_t* __restrict__ cursors, uint32_t* __restrict__ queues, float* __restrict__ dtable) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  const int level = level0 + blockIdx.y;
  const int B = 1 << (grid.log2_table_size - slice_log2);
  const bool combine = (combine_mask >> level) & 1u;  // workgroup-uniform
  // layout: [vals: 2 x slots floats][keys: slots][cnt: B][base: B]  (vals/keys only when any level combines)
  const int table_words = combine_mask != 0u ? 3 * kCombineSlots : 0;
  float* vals = reinterpret_cast<float*>(lds_u);
  uint32_t* keys = lds_u + 2 * kCombineSlots;
  uint32_t* cnt = lds_u + table_words;  // [B] updates of this workgroup per tile
  uint32_t* base = cnt + B;             // [B] reserved queue offset per tile
  for (int t = threadIdx.x; t < B; t += kBinThreads) cnt[t] = 0;
  if (combine) {
#pragma unroll
    for (int i = 0; i < kCombinePerThread; ++i) {
      const int sl = threadIdx.x + i * kBinThreads;
      keys[sl] = kEmptyKey;
      vals[2 * sl] = 0.0f;
      vals[2 * sl + 1] = 0.0f;
    }
  }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBinThreads + threadIdx.x;
  bool active = p < M;
  float g0 = 0.f, g1 = 0.f;
  if (active) {
    const float* gptr = denc + p * stride_p + (int64_t)(2 * level) * stride_k;
    g0 = gptr[0];
    g1 = gptr[stride_k];
    active = !(g0 == 0.0f && g1 == 0.0f);
  }
  uint32_t idx[8], rank[8];
  float v0[8], v1[8];
  Cell c;
#pragma unroll
  for (int a = 0; a < 3; ++a) { c.lo[a] = 0; c.hi[a] = 0; c.w[a] = 0.f; }
  if (active) {
    float x, y, z;
    load_position(P, p, x, y, z);
    (void)normalise_position(transform, box, x, y, z);
    c = locate_cell(x, y, z, grid.scalings[level]);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float bz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
    const float by = (k & 2) ? c.w[1] : 1.0f - c.w[1];
    const float bx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
    v0[k] = active ? ((g0 * bz) * by) * bx : 0.0f;
    v1[k] = active ? ((g1 * bz) * by) * bx : 0.0f;
  }
  bool emit = active;
  if ((merge_mask >> level) & 1u) {  // wave-uniform
    const int lane = threadIdx.x & 63;
    bool same = active && lane > 0 && __shfl_up((int)active, 1) != 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      same = same && (__shfl_up(c.lo[a], 1) == c.lo[a]) && (__shfl_up(c.hi[a], 1) == c.hi[a]);
    }
    const bool head = !same;
    bool f = head;  // segmented inclusive scan: (v, f) o (v', f') = (f' ? v' : v + v', f | f')
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const bool take = (lane >= d) && !f;
      const int fp = __shfl_up((int)f, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float a0 = __shfl_up(v0[k], d), a1 = __shfl_up(v1[k], d);
        if (take) { v0[k] += a0; v1[k] += a1; }
      }
      if (take) f = fp != 0;
    }
    const bool next_head = (lane == 63) || (__shfl_down((int)head, 1) != 0);
    emit = active && next_head;  // the last lane of a run carries the run's sums
  }
  uint32_t direct = 0u;  // bit k: corner k leaves this thread as its own record
  if (emit) {
    const uint32_t mask = (1u << grid.log2_table_size) - 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      idx[k] = corner_index(c, k, mask);
      bool placed = false;
      if (combine) {
        uint32_t h = (idx[k] * 0x9E3779B1u) >> (32 - kCombineBits);
        for (int probe = 0; probe < 4 && !placed; ++probe) {
          const uint32_t prev = atomicCAS(keys + h, kEmptyKey, idx[k]);
          if (prev == kEmptyKey || prev == idx[k]) {
            lds_add_pair(vals + 2 * h, v0[k], v1[k]);
            placed = true;
          } else {
            h = (h + 1) & (kCombineSlots - 1);
          }
        }
      }
      if (!placed) {
        direct |= 1u << k;
        rank[k] = atomicAdd(cnt + (idx[k] >> slice_log2), 1u);  // ds_add_rtn_u32
      }
    }
  }
  uint32_t skey[kCombinePerThread], srank[kCombinePerThread];
  if (combine) {
    __syncthreads();  // all sums of the workgroup are in the table
#pragma unroll
    for (int i = 0; i < kCombinePerThread; ++i) {
      skey[i] = keys[threadIdx.x + i * kBinThreads];
      srank[i] = 0u;
      if (skey[i] != kEmptyKey) srank[i] = atomicAdd(cnt + (skey[i] >> slice_log2), 1u);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < B; t += kBinThreads) {
    const uint32_t n = cnt[t];
    base[t] = n ? atomicAdd(cursors + (size_t)level * B + t, n) : 0u;
  }
  __syncthreads();
  const uint32_t local_mask = (1u << slice_log2) - 1u;
  auto put = [&](uint32_t index, uint32_t rk, float a0, float a1) {
    const uint32_t bin = index >> slice_log2;
    const uint32_t pos = base[bin] + rk;
    if (pos < cap) {  // one 16-B record = one global_store_dwordx4
      uint4* q = reinterpret_cast<uint4*>(queues) + (((size_t)level * B + bin) * cap + pos);
      *q = make_uint4(index & local_mask, __float_as_uint(a0), __float_as_uint(a1), 0u);
    } else {  // queue full (a very hot cell): direct atomics keep the result exact
      float* t = dtable + ((((size_t)level << grid.log2_table_size) + index) << 1);
      unsafeAtomicAdd(t + 0, a0);
      unsafeAtomicAdd(t + 1, a1);
    }
  };
  if (direct != 0u) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((direct >> k) & 1u) put(idx[k], rank[k], v0[k], v1[k]);
  }
  if (combine) {
#pragma unroll
    for (int i = 0; i < kCombinePerThread; ++i) {
      const int sl = threadIdx.x + i * kBinThreads;
      if (skey[i] != kEmptyKey) put(skey[i], srank[i], vals[2 * sl], vals[2 * sl + 1]);
    }
  }
}

template <bool kPipe>
__global__ void hash_bwd_apply_kernel(nsamd_grid grid, int slice_log2, uint32_t cap, int level0,
                                      const uint32_t* __restrict__ cursors, const uint32_t* __restrict__ queues,
                                      float* __restrict__ dtable) {
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
  const uint4* q = reinterpret_cast<const uint4*>(queues) + ((size_t)level * B + bin) * cap;
  // One workgroup per CU (the tile fills the LDS), so memory-level parallelism has to come from each thread: keep 8
  // independent 16-B queue loads in flight before touching the LDS (measured: 285 -> see profiles/ us per launch).
  constexpr int kU = 8;
  uint32_t e = threadIdx.x;
  const uint32_t stride = blockDim.x;
  if (kPipe) {
    // software pipeline: the next batch of queue loads is in flight while this batch goes through the LDS
    uint4 cur[kU];
    bool have = e + (kU - 1) * stride < n;
    if (have) {
#pragma unroll
      for (int u = 0; u < kU; ++u) cur[u] = q[e + u * stride];
    }
    while (have) {
      const uint32_t en = e + kU * stride;
      const bool more = en + (kU - 1) * stride < n;
      uint4 nxt[kU];
      if (more) {
#pragma unroll
        for (int u = 0; u < kU; ++u) nxt[u] = q[en + u * stride];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        lds_add_pair(acc + 2 * cur[u].x, __uint_as_float(cur[u].y), __uint_as_float(cur[u].z));
      if (more) {
#pragma unroll
        for (int u = 0; u < kU; ++u) cur[u] = nxt[u];
      }
      e = en;
      have = more;
    }
  }
  for (; e + (kU - 1) * stride < n; e += kU * stride) {
    uint4 r[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) r[u] = q[e + u * stride];
    // (Measured: issuing the 8 reads, then the 8 CAS attempts, widens the read->CAS window enough that races between the
    // waves on small hot tiles make it 2.4x SLOWER for the proposal tables — every lost race pays the divergent
    // ds_add_f32 path. Keep read and CAS of a record adjacent.)
#pragma unroll
    for (int u = 0; u < kU; ++u) lds_add_pair(acc + 2 * r[u].x, __uint_as_float(r[u].y), __uint_as_float(r[u].z));
  }
  for (; e < n; e += stride) {
    const uint4 r = q[e];
    lds_add_pair(acc + 2 * r.x, __uint_as_float(r.y), __uint_as_float(r.z));
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(
      dtable + ((((size_t)level << grid.log2_table_size) + ((size_t)bin << slice_log2)) << 1));
  const float4* a4 = reinterpret_cast<const float4*>(acc);
  for (int i = threadIdx.x; i < entries / 2; i += blockDim.x) {  // sole owner of the tile: plain read-modify-write
    float4 o = out[i];
    const float4 a = a4[i];
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    out[i] = o;
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

// tiles (= bins x levels) the binned scatter aims for; tunable for experiments through NSAMD_SCATTER_TILES
static int scatter_target_tiles() {
  static int cached = 0;
  if (cached == 0) {
    const char* e = getenv("NSAMD_SCATTER_TILES");
    cached = (e != nullptr && atoi(e) >= 64) ? atoi(e) : 512;
  }
  return cached;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e != nullptr ? atoi(e) : dflt;
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

extern "C" int nsamd_hashgrid_encode_bwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                         const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                         int64_t stride_k, float* dtable, float* dpositions, float* workspace,
                                         int64_t workspace_floats, nsamd_stream_t stream) {
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
  if (dtable != nullptr && M < 8192) {
    // small batches: direct fire-and-forget atomics (zeroing / writing whole tiles would dominate)
    dim3 g((unsigned)nb, (unsigned)grid.num_levels);
    hash_encode_bwd_table_kernel<<<g, kHashBlock, 0, (hipStream_t)stream>>>(pts, M, transform, aabb, grid, denc,
                                                                             stride_p, stride_k, dtable);
    NSAMD_CHECK_LAUNCH();
  } else if (dtable != nullptr && [&]() -> bool {
               // binned path: tile size chosen so that (tiles = bins x levels) >= 512 fills the chip
               int bits = 0;
               while ((grid.num_levels << bits) < scatter_target_tiles()) ++bits;
               int sl = grid.log2_table_size - bits;
               sl = sl > kSliceLog2Max ? kSliceLog2Max : (sl < 8 ? 8 : sl);
               if (sl > grid.log2_table_size) sl = grid.log2_table_size;
               const int64_t B = (int64_t)1 << (grid.log2_table_size - sl);
               if (workspace == nullptr || B > kMaxBins) return false;
               const int64_t tiles = B * grid.num_levels;
               const int64_t cap = (workspace_floats - tiles - 4) / (4 * tiles);
               const int64_t expect = (8 * M + B - 1) / B;  // uniform hashing: updates per tile
               return cap >= expect + expect / 4 && cap < 0x7fffffffLL;
             }()) {
    int bits = 0;
    while ((grid.num_levels << bits) < scatter_target_tiles()) ++bits;
    int sl = grid.log2_table_size - bits;
    sl = sl > kSliceLog2Max ? kSliceLog2Max : (sl < 8 ? 8 : sl);
    if (sl > grid.log2_table_size) sl = grid.log2_table_size;
    const int B = 1 << (grid.log2_table_size - sl);
    const int64_t tiles = (int64_t)B * grid.num_levels;
    const uint32_t cap = (uint32_t)((workspace_floats - tiles - 4) / (4 * tiles));
    uint32_t* cursors = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* queues = cursors + ((tiles + 3) & ~(int64_t)3);  // 16-B aligned records
    hipStream_t st = (hipStream_t)stream;
    static const int groups_env = env_int("NSAMD_SCATTER_GROUPS", 1);
    static const float merge_env = (float)env_int("NSAMD_SCATTER_MERGE_X4", 16) * 0.25f;
    static const int pipe_env = env_int("NSAMD_SCATTER_PIPE", 1);
    const int groups = groups_env < 1 ? 1 : (groups_env > grid.num_levels ? grid.num_levels : groups_env);
    const int per_group = (grid.num_levels + groups - 1) / groups;
    // merge runs of samples that share a cell where the cell is wider than ~4 sample spacings (ray mode only: the
    // lanes of a wave are then consecutive samples of one ray)
    uint32_t merge_mask = 0;
    if (pts.positions == nullptr)
      for (int l = 0; l < grid.num_levels; ++l)
        if (grid.scalings[l] < merge_env * (float)pts.samples_per_ray) merge_mask |= 1u << l;
    // per-workgroup combining pays where a workgroup's updates hit few distinct entries: coarse levels
    static const int combine_env = env_int("NSAMD_SCATTER_COMBINE_RES", 64);
    uint32_t combine_mask = 0;
    for (int l = 0; l < grid.num_levels; ++l)
      if (grid.scalings[l] < (float)combine_env) combine_mask |= 1u << l;
    const size_t bin_lds = sizeof(uint32_t) * (2 * (size_t)B + (combine_mask ? 3 * kCombineSlots : 0));
    static bool attr2 = false;
    if (!attr2) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_bwd_apply_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(float) << kSliceLog2Max) !=
              hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_bwd_apply_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * sizeof(float) << kSliceLog2Max) !=
              hipSuccess)
        return NSAMD_ERR_LAUNCH;
      attr2 = true;
    }
    const unsigned threads = sl > 11 ? 1024u : 256u;
    static const int only_env = env_int("NSAMD_SCATTER_ONLY_LEVEL", -1);  // diagnostics: a single level
    for (int l0 = only_env >= 0 ? only_env : 0; l0 < (only_env >= 0 ? only_env + 1 : grid.num_levels); l0 += per_group) {
      const int nl = only_env >= 0 ? 1 : (grid.num_levels - l0 < per_group ? grid.num_levels - l0 : per_group);
      dim3 g1((unsigned)((M + kBinThreads - 1) / kBinThreads), (unsigned)nl);
      hash_bwd_bin_kernel<<<g1, kBinThreads, bin_lds, st>>>(pts, M, transform, aabb, grid, denc, stride_p, stride_k, sl,
                                                            cap, merge_mask, combine_mask, l0, cursors, queues, dtable);
      NSAMD_CHECK_LAUNCH();
      dim3 g2((unsigned)B, (unsigned)nl);
      if (pipe_env)
        hash_bwd_apply_kernel<true><<<g2, threads, sizeof(float) * 2 * ((size_t)1 << sl), st>>>(
            grid, sl, cap, l0, cursors, queues, dtable);
      else
        hash_bwd_apply_kernel<false><<<g2, threads, sizeof(float) * 2 * ((size_t)1 << sl), st>>>(
            grid, sl, cap, l0, cursors, queues, dtable);
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
