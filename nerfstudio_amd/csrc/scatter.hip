// Table-gradient scatter of the hash encoding for gfx950 (backward of HashEncoding.pytorch_fwd,
// /root/reference/nerfstudio/field_components/encodings.py:417-458 — autograd's index_put accumulate into the dense
// [L*T, 2] gradient). HBM-bound byte work: 8 corners x 16 B read-modify-write per (point, level) algorithmically.
//
// fp32 global atomics retire at ~20 G lane-ops/s on this part (profiles/r01_probe_scatter*.log) — 2.9 ms for the
// nerfacto main table — so the scatter is two passes over 16-B records with NO float atomics anywhere:
//
//  PASS 1 (route)  derives every corner update once and appends a record to the queue of the (level, tile) it hashes
//    into. Records are x-PAIRS: idx = x ^ y*P1 ^ z*P2 differs only in low bits between x and x+1, so both
//    x-neighbours of a cell edge share a tile and everything but the x weight: (g0*bz*by, g1*bz*by, wx, ia|ib<<14|flag).
//    Fine levels (scatter_route_fine_kernel): a tile's queue starts with one STATIC segment per pass-1 workgroup, so a
//    record's slot is  segment base + ds_add_rtn rank  — no reservation round trip to global memory, one barrier, the
//    store leaves as soon as the rank is back. Only what overflows a segment (non-uniform levels) takes the classic
//    route: one returning global atomic per (workgroup, tile) into the tile's dynamic area, records recomputed in a
//    second sweep. Coarse levels (scatter_route_runs_kernel, cell wider than the sample spacing): a thread walks 4
//    consecutive samples and sums the corner contributions in registers while the cell stays the same (fixed order),
//    then count -> reserve -> emit in two sweeps.
//  PASS 2 (apply)  one workgroup per tile accumulates its queue in LDS and stores (write-only call) or adds the tile
//    with coalesced accesses.
//
// Determinism: pass 2 accumulates in 64-bit fixed point (scatter.h) with ds_add_u64 — native, fire-and-forget, 3.0
// lane-ops/clk/CU with divergent addresses and no slower on one hot address (the float CAS loop of round 1: 2.2, and
// it collapsed on hot coarse cells). Integer addition is associative, so queue order, the static/dynamic/spill
// routing and the atomics' order do not reach the result: two runs on the same inputs give the same bits. Records that
// find no room in their tile go to a spill list; pass 2 folds the first kSpillFold of them into their tiles (still
// exact and order-free), the finish kernel applies any rest with float atomics and counts them (hdr[kHdrEvtUnordered]).
#include <stdlib.h>

#include <type_traits>

#include "field_reduce.h"
#include "scatter.h"

NSAMD_PROBE_DEFINE(scatter)

namespace nsamd {

// Gated calls (nsamd_hashgrid_encode_bwd_gated): `gate` points at the flag nsamd_weights_bwd_gate raises when any ray
// of the level carries gradient. While it is clear every value of `denc` would be an exact zero (it is not even
// written), the accumulating scatter adds nothing, and all of its kernels return at once.
__device__ __forceinline__ bool gate_is_clear(const uint32_t* gate) {
  return gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

constexpr int kRunLen = 4;        // consecutive samples per thread in the run kernel
constexpr int kRunThreads = 256;  // -> 1024 points per workgroup
constexpr int kMaxLog2Bins = 10;  // pass-1 LDS counters: 3 x 4 levels x bins x 4 B <= 48 KiB

// ---- pass 1, fine levels -------------------------------------------------------------------------------------------
// kThreads x kPts points per workgroup, kLevels levels per thread (position computed once, 4 * kLevels * kPts
// independent record chains per thread to cover the load -> LDS rank -> store latencies).
template <int kThreads, int kPts, int kLevels>
__global__ __launch_bounds__(kThreads) void scatter_route_fine_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, ScatterGeom G, LevelList levels, ScatterBufs buf,
    const uint32_t* __restrict__ gate, const uint8_t* __restrict__ ray_mask) {
  static_assert(kPts * kLevels * 4 <= 32, "overflow mask is 32 bits");
  if (gate_is_clear(gate)) return;  // gated call with no gradient anywhere: nothing to route (see scatter_launch)
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  const int B = 1 << G.log2_bins;
  uint32_t* cnt = lds_u;               // [kLevels][B] records of this workgroup per tile
  uint32_t* cnt2 = cnt + kLevels * B;  // [kLevels][B] ranks inside the dynamic reservation (second sweep)
  uint32_t* base = cnt2 + kLevels * B; // [kLevels][B] start of the dynamic reservation
  uint32_t* lmax = base + kLevels * B; // [kLevels] max |gradient| bits
  PROBE_STAMP(0, 0);
  for (int t = threadIdx.x; t < 2 * kLevels * B; t += kThreads) cnt[t] = 0u;
  if (threadIdx.x < kLevels) lmax[threadIdx.x] = 0u;
  const int first = blockIdx.y * kLevels;
  int lvl[kLevels];
#pragma unroll
  for (int i = 0; i < kLevels; ++i) lvl[i] = first + i < levels.count ? (int)levels.level[first + i] : -1;
  float g0[kPts][kLevels], g1[kPts][kLevels];
  float x[kPts], y[kPts], z[kPts];
  bool inside[kPts];
  // Every load of the prologue — the ray's mask byte, the 2 * kLevels gradients, the position's inputs — is UNCONDITIONAL (at a
  // clamped point, a level-0 address for an absent level, a dummy byte without a mask) and issued before the first is used;
  // what must not count is zeroed by a select afterwards. Under their predicates (`inside && lvl >= 0`, the mask) each load
  // sat in a branch of its own and was waited for before the next was issued: mask, gradients, bin edges, origin / direction
  // were four memory round trips in a row per workgroup (read off the ISA), which is all a sparse gated call consists of.
  uint8_t mask_raw[kPts];
  int64_t pcl[kPts];
  // (the mask bytes FIRST, alone: a workgroup none of whose rays carries gradient — most of them while the interlevel loss
  //  reaches few rays — leaves after this one small round trip, before it has asked for a single gradient or position)
#pragma unroll
  for (int j = 0; j < kPts; ++j) {
    const int64_t p = ((int64_t)blockIdx.x * kPts + j) * kThreads + threadIdx.x;
    inside[j] = p < M;
    pcl[j] = inside[j] ? p : M - 1;
    // per-ray mask of a gated call: samples of rays without gradient are exact zeros that were never written
    mask_raw[j] = ray_mask != nullptr ? ray_mask[pcl[j] / P.samples_per_ray] : (uint8_t)1;
  }
  if (ray_mask != nullptr) {
    bool marked = false;
#pragma unroll
    for (int j = 0; j < kPts; ++j) marked = marked || (inside[j] && mask_raw[j] != 0);
    if (!__syncthreads_or(marked)) {  // nothing to route: publish the zero segment counts of this workgroup's levels and leave
      for (int t = threadIdx.x; t < kLevels * B; t += kThreads) {
        const int i = t >> G.log2_bins;
        if (lvl[i] < 0) continue;
        const uint32_t tile = ((uint32_t)lvl[i] << G.log2_bins) + (uint32_t)(t & (B - 1));
        buf.counts[(size_t)tile * G.segs + blockIdx.x] = 0u;
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < kPts; ++j) {
    const int64_t pc = pcl[j];
#pragma unroll
    for (int i = 0; i < kLevels; ++i) {
      const float* gptr = denc + pc * stride_p + (int64_t)(2 * (lvl[i] >= 0 ? lvl[i] : 0)) * stride_k;
      g0[j][i] = gptr[0];
      g1[j][i] = gptr[stride_k];
    }
  }
  load_positions_burst<kPts>(P, pcl, x, y, z);
#pragma unroll
  for (int j = 0; j < kPts; ++j) {
    if (ray_mask != nullptr) inside[j] = inside[j] && mask_raw[j] != 0;
#pragma unroll
    for (int i = 0; i < kLevels; ++i) {
      const bool live = inside[j] && lvl[i] >= 0;
      g0[j][i] = live ? g0[j][i] : 0.0f;
      g1[j][i] = live ? g1[j][i] : 0.0f;
    }
    if (inside[j]) {
      (void)normalise_position(transform, box, x[j], y[j], z[j]);
    } else {
      x[j] = y[j] = z[j] = 0.0f;
    }
  }
  // A workgroup whose points carry no gradient at all (proposal levels: the interlevel loss reaches few samples,
  // profiles/r02_study_proposal_sparsity.txt) has nothing to route: it skips the sweep and only publishes its zero
  // segment counts. (`x != 0` is true for NaN: a non-finite gradient is never skipped.) The test rides on the barrier
  // that was here anyway and adds no dependence to the loads above.
  bool carries = false;
#pragma unroll
  for (int j = 0; j < kPts; ++j)
#pragma unroll
    for (int i = 0; i < kLevels; ++i) carries = carries || g0[j][i] != 0.0f || g1[j][i] != 0.0f;
  const int any_gradient = __syncthreads_or(carries);
  PROBE_STAMP(0, 1);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const int sl = G.slice_log2;
  const uint32_t local_mask = (1u << sl) - 1u;
  const uint32_t C = G.seg_cap;
  const uint32_t static_end = G.segs * C;
  uint32_t over = 0u;  // bit ((j * kLevels + i) * 4 + q): the record found its static segment full

  auto sweep = [&](auto tag) {
    constexpr int SW = decltype(tag)::value;
#pragma unroll
    for (int j = 0; j < kPts; ++j) {
#pragma unroll
      for (int i = 0; i < kLevels; ++i) {
        const int slot = (j * kLevels + i) * 4;
        if (lvl[i] < 0 || !inside[j] || (g0[j][i] == 0.0f && g1[j][i] == 0.0f)) continue;  // adding zero is a no-op
        if (SW == 1 && ((over >> slot) & 0xfu) == 0u) continue;
        const Cell c = locate_cell(x[j], y[j], z[j], grid.scalings[lvl[i]]);
        if (SW == 0) {  // integer compare of |bits|: a NaN or Inf wins and marks the level non-finite
          const uint32_t b0 = __float_as_uint(g0[j][i]) & 0x7fffffffu, b1 = __float_as_uint(g1[j][i]) & 0x7fffffffu;
          atomicMax(lmax + i, b0 > b1 ? b0 : b1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (SW == 1 && !((over >> (slot + q)) & 1u)) continue;
          const PairHash h = pair_hash(c, q, mask);
          const uint32_t bin = h.ia >> sl;
          const uint32_t tile = ((uint32_t)lvl[i] << G.log2_bins) + bin;
          const uint32_t Q = G.level_cap[lvl[i]];
          uint4* const queue = buf.queues + ((size_t)G.level_off[lvl[i]] + (size_t)bin * Q);
          // autograd order ((g * wz) * wy) * wx; the x factor is applied by pass 2
          const float bz = (q & 2) ? c.w[2] : 1.0f - c.w[2];
          const float by = (q & 1) ? c.w[1] : 1.0f - c.w[1];
          const float a0 = (g0[j][i] * bz) * by, a1 = (g1[j][i] * bz) * by;
          if ((h.ib >> sl) != bin) {
            // the pair straddles two tiles (needs a carry past bit slice_log2: only when res >= 2^slice_log2, which the
            // host avoids whenever the table allows): two single records through the spill list, in the second sweep
            if (SW == 0) {
              over |= 1u << (slot + q);
            } else {
              spill_append(buf, G.spill_cap, tile,
                           make_uint4(__float_as_uint(a0 * (1.0f - c.w[0])), __float_as_uint(a1 * (1.0f - c.w[0])), 0u,
                                      h.ia & local_mask));
              spill_append(buf, G.spill_cap, ((uint32_t)lvl[i] << G.log2_bins) + (h.ib >> sl),
                           make_uint4(__float_as_uint(a0 * c.w[0]), __float_as_uint(a1 * c.w[0]), 0u, h.ib & local_mask));
            }
            continue;
          }
          const uint4 rec = make_uint4(__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(c.w[0]),
                                       (h.ia & local_mask) | ((h.ib & local_mask) << 14) | 0x80000000u);
          if (SW == 0) {
            const uint32_t rank = atomicAdd(cnt + (i << G.log2_bins) + bin, 1u);  // ds_add_rtn_u32
            if (rank < C) rec_store(queue + (blockIdx.x * C + rank), rec);
            else over |= 1u << (slot + q);
          } else {
            const uint32_t pos = base[(i << G.log2_bins) + bin] + atomicAdd(cnt2 + (i << G.log2_bins) + bin, 1u);
            if (pos < Q - static_end) rec_store(queue + (static_end + pos), rec);
            else spill_append(buf, G.spill_cap, tile, rec);
          }
        }
      }
    }
  };

  if (any_gradient) sweep(std::integral_constant<int, 0>{});
  PROBE_STAMP(0, 2);
  const int any_over = __syncthreads_or(over != 0u);
  PROBE_STAMP(0, 3);
  for (int t = threadIdx.x; t < kLevels * B; t += kThreads) {
    const int i = t >> G.log2_bins;
    const int level = first + i < levels.count ? (int)levels.level[first + i] : -1;
    if (level < 0) continue;
    const uint32_t n = cnt[t];
    const uint32_t tile = ((uint32_t)level << G.log2_bins) + (uint32_t)(t & (B - 1));
    buf.counts[(size_t)tile * G.segs + blockIdx.x] = n < C ? n : C;
    if (any_over) base[t] = n > C ? atomicAdd(buf.dyn_cursor + tile, n - C) : 0u;
  }
  if (threadIdx.x < kLevels) {
    const int level = first + (int)threadIdx.x < levels.count ? (int)levels.level[first + threadIdx.x] : -1;
    if (level >= 0 && lmax[threadIdx.x] != 0u) atomicMax(buf.hdr + level, lmax[threadIdx.x]);
  }
  PROBE_STAMP(0, 4);
  if (!any_over) return;
  __syncthreads();
  sweep(std::integral_constant<int, 1>{});
  PROBE_STAMP(0, 5);
}

// ---- pass 1, coarse levels -----------------------------------------------------------------------------------------
// Every thread walks kRunLen CONSECUTIVE samples and merges those that stay in one cell (a run): per-corner sums in
// registers, in sample order. A run of one sample leaves as 4 x-pair records, a longer one as 8 single records
// (value pair + local index). Sweep 0 counts per tile, the workgroup reserves its share of every tile's queue with one
// returning atomic, sweep 1 recomputes the runs and stores — nothing is kept in registers or LDS between the sweeps.
template <int kLevels>
__device__ __forceinline__ void scatter_route_runs_body(
    const nsamd_points& P, int64_t M, int transform, const nsamd_aabb& box, const nsamd_grid& grid,
    const float* __restrict__ denc, int64_t stride_p, int64_t stride_k, const ScatterGeom& G, const LevelList& levels,
    const ScatterBufs& buf, const uint32_t* __restrict__ gate, const uint8_t* __restrict__ ray_mask) {
  if (gate_is_clear(gate)) return;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  const int B = 1 << G.log2_bins;
  uint32_t* cnt = lds_u;
  uint32_t* cnt2 = cnt + kLevels * B;
  uint32_t* base = cnt2 + kLevels * B;
  uint32_t* lmax = base + kLevels * B;
  PROBE_STAMP(0, 20);
  for (int t = threadIdx.x; t < 2 * kLevels * B; t += kRunThreads) cnt[t] = 0u;
  if (threadIdx.x < kLevels) lmax[threadIdx.x] = 0u;
  const int first = blockIdx.y * kLevels;
  int lvl[kLevels];
#pragma unroll
  for (int i = 0; i < kLevels; ++i) lvl[i] = first + i < levels.count ? (int)levels.level[first + i] : -1;
  const int64_t p0 = ((int64_t)blockIdx.x * kRunThreads + threadIdx.x) * kRunLen;
  float g0[kRunLen][kLevels], g1[kRunLen][kLevels];
  float px[kRunLen], py[kRunLen], pz[kRunLen];
  bool act[kRunLen];
  // (unconditional loads at clamped points, all issued before the first is used, selects afterwards: see the fine kernel —
  //  here the predicated form was one round trip per sample for the mask and gradients and two more per sample for the position)
  uint8_t mask_raw[kRunLen];
  int64_t pcl[kRunLen];
#pragma unroll
  for (int s = 0; s < kRunLen; ++s) {  // (the mask bytes first, alone: see the fine kernel)
    act[s] = p0 + s < M;
    pcl[s] = act[s] ? p0 + s : M - 1;
    mask_raw[s] = ray_mask != nullptr ? ray_mask[pcl[s] / P.samples_per_ray] : (uint8_t)1;
  }
  if (ray_mask != nullptr) {
    bool marked = false;
#pragma unroll
    for (int s = 0; s < kRunLen; ++s) marked = marked || (act[s] && mask_raw[s] != 0);
    if (!__syncthreads_or(marked)) return;  // (this kernel keeps no per-workgroup state in the workspace)
  }
#pragma unroll
  for (int s = 0; s < kRunLen; ++s) {
    const int64_t pc = pcl[s];
#pragma unroll
    for (int i = 0; i < kLevels; ++i) {
      const float* gptr = denc + pc * stride_p + (int64_t)(2 * (lvl[i] >= 0 ? lvl[i] : 0)) * stride_k;
      g0[s][i] = gptr[0];
      g1[s][i] = gptr[stride_k];
    }
  }
  load_positions_burst<kRunLen>(P, pcl, px, py, pz);
#pragma unroll
  for (int s = 0; s < kRunLen; ++s) {
    if (ray_mask != nullptr) act[s] = act[s] && mask_raw[s] != 0;
#pragma unroll
    for (int i = 0; i < kLevels; ++i) {
      const bool live = act[s] && lvl[i] >= 0;
      g0[s][i] = live ? g0[s][i] : 0.0f;
      g1[s][i] = live ? g1[s][i] : 0.0f;
    }
    if (act[s]) {
      (void)normalise_position(transform, box, px[s], py[s], pz[s]);
    } else {
      px[s] = py[s] = pz[s] = 0.0f;
    }
  }
  // no gradient anywhere in this workgroup's samples: nothing to count, reserve or emit — and this kernel keeps no
  // per-workgroup state in the workspace, so the workgroup may simply leave (the test rides on the barrier that was
  // here anyway; `x != 0` is true for NaN)
  {
    bool carries = false;
#pragma unroll
    for (int s = 0; s < kRunLen; ++s)
#pragma unroll
      for (int i = 0; i < kLevels; ++i) carries = carries || g0[s][i] != 0.0f || g1[s][i] != 0.0f;
    if (!__syncthreads_or(carries)) return;
  }
  PROBE_STAMP(0, 21);
  const uint32_t mask = (1u << grid.log2_table_size) - 1u;
  const int sl = G.slice_log2;
  const uint32_t local_mask = (1u << sl) - 1u;

  auto sweep = [&](auto tag) {
    constexpr int SW = decltype(tag)::value;
    auto emit = [&](int i, uint32_t tile_bin, uint4 rec) {
      const uint32_t tile = ((uint32_t)lvl[i] << G.log2_bins) + tile_bin;
      const uint32_t Q = G.level_cap[lvl[i]];
      if (SW == 0) {
        atomicAdd(cnt + (i << G.log2_bins) + tile_bin, 1u);
      } else {
        const uint32_t pos = base[(i << G.log2_bins) + tile_bin] + atomicAdd(cnt2 + (i << G.log2_bins) + tile_bin, 1u);
        if (pos < Q) rec_store(buf.queues + ((size_t)G.level_off[lvl[i]] + (size_t)tile_bin * Q + pos), rec);
        else spill_append(buf, G.spill_cap, tile, rec);
      }
    };
#pragma unroll
    for (int i = 0; i < kLevels; ++i) {
      if (lvl[i] < 0) continue;
      const float scale = grid.scalings[lvl[i]];
      Cell cur{};
      float a0[8], a1[8];       // per-corner sums of the open run (valid when len >= 2)
      float f0 = 0.f, f1 = 0.f; // gradient of the run's first sample (len == 1: pair records)
      int len = 0;
      uint32_t gmax = 0u;  // max |gradient| bits (integer compare: NaN / Inf win)
#pragma unroll
      for (int s = 0; s <= kRunLen; ++s) {
        bool live = false;
        Cell c = cur;
        if (s < kRunLen) {
          live = (p0 + s < M) && !(g0[s][i] == 0.0f && g1[s][i] == 0.0f);
          if (live) c = locate_cell(px[s], py[s], pz[s], scale);
        }
        const bool same = len > 0 && live && c.lo[0] == cur.lo[0] && c.lo[1] == cur.lo[1] && c.lo[2] == cur.lo[2] &&
                          c.hi[0] == cur.hi[0] && c.hi[1] == cur.hi[1] && c.hi[2] == cur.hi[2];
        if (len > 0 && (s == kRunLen || (live && !same))) {  // the run ends
          if (len == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const PairHash h = pair_hash(cur, q, mask);
              const float bz = (q & 2) ? cur.w[2] : 1.0f - cur.w[2];
              const float by = (q & 1) ? cur.w[1] : 1.0f - cur.w[1];
              const float v0 = (f0 * bz) * by, v1 = (f1 * bz) * by;
              if ((h.ib >> sl) == (h.ia >> sl)) {
                emit(i, h.ia >> sl,
                     make_uint4(__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(cur.w[0]),
                                (h.ia & local_mask) | ((h.ib & local_mask) << 14) | 0x80000000u));
              } else {  // straddling pair: two singles
                emit(i, h.ia >> sl, make_uint4(__float_as_uint(v0 * (1.0f - cur.w[0])),
                                               __float_as_uint(v1 * (1.0f - cur.w[0])), 0u, h.ia & local_mask));
                emit(i, h.ib >> sl, make_uint4(__float_as_uint(v0 * cur.w[0]), __float_as_uint(v1 * cur.w[0]), 0u,
                                               h.ib & local_mask));
              }
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t idx = corner_index(cur, k, mask);
              emit(i, idx >> sl, make_uint4(__float_as_uint(a0[k]), __float_as_uint(a1[k]), 0u, idx & local_mask));
            }
          }
          len = 0;
        }
        if (s < kRunLen && live) {
          if (SW == 0) {
            const uint32_t b0 = __float_as_uint(g0[s][i]) & 0x7fffffffu, b1 = __float_as_uint(g1[s][i]) & 0x7fffffffu;
            gmax = max(gmax, max(b0, b1));
          }
          if (same) {
            if (len == 1) {  // second sample of the run: open the per-corner sums with the first one (cell weights of cur)
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float bz = (k & 4) ? cur.w[2] : 1.0f - cur.w[2];
                const float by = (k & 2) ? cur.w[1] : 1.0f - cur.w[1];
                const float bx = (k & 1) ? cur.w[0] : 1.0f - cur.w[0];
                a0[k] = ((f0 * bz) * by) * bx;
                a1[k] = ((f1 * bz) * by) * bx;
              }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float bz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
              const float by = (k & 2) ? c.w[1] : 1.0f - c.w[1];
              const float bx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
              a0[k] += ((g0[s][i] * bz) * by) * bx;
              a1[k] += ((g1[s][i] * bz) * by) * bx;
            }
            len += 1;
          } else {
            cur = c;
            f0 = g0[s][i];
            f1 = g1[s][i];
            len = 1;
          }
        }
      }
      // a merged record carries up to kRunLen gradients: the scale bound accounts for that
      // (exponent + 2 = x kRunLen; saturates into the non-finite range only for gradients beyond 2^125)
      if (SW == 0 && gmax != 0u) atomicMax(lmax + i, min(gmax + (2u << 23), 0x7fc00000u));
    }
  };

  sweep(std::integral_constant<int, 0>{});
  PROBE_STAMP(0, 22);
  __syncthreads();
  for (int t = threadIdx.x; t < kLevels * B; t += kRunThreads) {
    const int i = t >> G.log2_bins;
    const int level = first + i < levels.count ? (int)levels.level[first + i] : -1;
    if (level < 0) continue;
    const uint32_t n = cnt[t];
    const uint32_t tile = ((uint32_t)level << G.log2_bins) + (uint32_t)(t & (B - 1));
    base[t] = n ? atomicAdd(buf.dyn_cursor + tile, n) : 0u;
  }
  if (threadIdx.x < kLevels) {
    const int level = first + (int)threadIdx.x < levels.count ? (int)levels.level[first + threadIdx.x] : -1;
    if (level >= 0 && lmax[threadIdx.x] != 0u) atomicMax(buf.hdr + level, lmax[threadIdx.x]);
  }
  __syncthreads();
  PROBE_STAMP(0, 23);
  sweep(std::integral_constant<int, 1>{});
  PROBE_STAMP(0, 24);
}

template <int kLevels>
__global__ __launch_bounds__(kRunThreads) void scatter_route_runs_kernel(
    nsamd_points P, int64_t M, int transform, nsamd_aabb box, nsamd_grid grid, const float* __restrict__ denc,
    int64_t stride_p, int64_t stride_k, ScatterGeom G, LevelList levels, ScatterBufs buf,
    const uint32_t* __restrict__ gate, const uint8_t* __restrict__ ray_mask) {
  scatter_route_runs_body<kLevels>(P, M, transform, box, grid, denc, stride_p, stride_k, G, levels, buf, gate, ray_mask);
}

// TWO independent scatters in one launch (the two proposal levels' table gradients on an update iteration: each of these
// kernels is as long as its slowest workgroup's chain of round trips, not as its work — side by side they cost one such chain).
// blockIdx.z selects the call; a call's own grid is a corner of the launch's.
struct RunsCall {
  nsamd_points P;
  int64_t M;
  int transform;
  nsamd_aabb box;
  nsamd_grid grid;
  const float* denc;
  int64_t stride_p, stride_k;
  ScatterGeom G;
  LevelList levels;
  ScatterBufs buf;
  const uint32_t* gate;
  const uint8_t* ray_mask;
  uint32_t grid_x, grid_y;
};

template <int kLevels>
__global__ __launch_bounds__(kRunThreads) void scatter_route_runs_pair_kernel(RunsCall a, RunsCall b) {
  if (blockIdx.z == 0) {
    if (blockIdx.x >= a.grid_x || blockIdx.y >= a.grid_y) return;
    scatter_route_runs_body<kLevels>(a.P, a.M, a.transform, a.box, a.grid, a.denc, a.stride_p, a.stride_k, a.G, a.levels, a.buf,
                                     a.gate, a.ray_mask);
  } else {
    if (blockIdx.x >= b.grid_x || blockIdx.y >= b.grid_y) return;
    scatter_route_runs_body<kLevels>(b.P, b.M, b.transform, b.box, b.grid, b.denc, b.stride_p, b.stride_k, b.G, b.levels, b.buf,
                                     b.gate, b.ray_mask);
  }
}

// ---- pass 2 ------------------------------------------------------------------------------------------------------
// One workgroup per (level, tile): static segments, dynamic area and the folded part of the spill list are summed into
// an LDS tile of 2 x int64 per entry; the finished tile is converted once and stored / added with coalesced accesses.
// kRider: the launch carries the main field's weight-gradient reduce along (its own instantiation: the rider's registers —
// 118 against 83 — must not cost the small-tile launches of the proposal levels their occupancy).
template <bool kRider>
__device__ __forceinline__ void scatter_apply_body(const nsamd_grid& grid, const ScatterGeom& G, const ScatterBufs& buf,
                                                   float* __restrict__ dtable, int overwrite,
                                                   const uint32_t* __restrict__ gate, const ReduceRider& rider) {
  if (gate_is_clear(gate)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];  // [entries][2]
  if (kRider && (int)blockIdx.y >= G.num_levels) {
    // RIDER workgroups (rows of the grid behind the levels): the main field's weight-gradient reduce, which needs nothing of
    // this pass and which this pass needs nothing of (field_reduce.h) — dispatched after the tiles, they fill the compute
    // units the last round of tiles leaves idle instead of being a launch of their own
    const int block = ((int)blockIdx.y - G.num_levels) * (int)gridDim.x + (int)blockIdx.x;
    if (block < rider.blocks)
      field_dw_reduce_body(reinterpret_cast<float*>(acc), block, rider.partials, rider.num_partials, rider.grads, rider.app_dim,
                           rider.app_rows, rider.cams, rider.num_rays, rider.tiles_per_ray);
    return;
  }
  const int bin = blockIdx.x, level = blockIdx.y;
  const uint32_t tile = ((uint32_t)level << G.log2_bins) + (uint32_t)bin;
  const int entries = 1 << G.slice_log2;
  PROBE_STAMP(0, 10);
  const bool coarse = (G.coarse_mask >> level) & 1u;
  const uint32_t Q = G.level_cap[level], C = G.seg_cap;
  // headroom of the fixed-point sums: <= 2 summands per record (a pair whose corners coincide) over the queue and the
  // folded spill records; one more bit for the sign
  const int headroom = 2 + (32 - __clz((int)(Q + kSpillFold - 1u)));
  const FixedScale fs = fixed_scale(buf.hdr[level], headroom);
  if (fs.empty && !overwrite) {
    // an accumulating call and nothing (or only flushed denormals) recorded on this level: the tile stays as it is
    if (threadIdx.x == 0) buf.dyn_cursor[tile] = 0u;
    return;
  }
  const uint32_t static_end = coarse ? 0u : G.segs * C;
  const uint32_t n_dyn = min(buf.dyn_cursor[tile], Q - static_end);
  const uint32_t n_spill = min(min(buf.hdr[kHdrSpillCount], G.spill_cap), kSpillFold);
  if (!overwrite && n_spill == 0u) {
    // Accumulating call: a tile nothing was routed into (the proposal levels while few rays carry gradient,
    // profiles/r03_proposal_sparsity.txt) keeps its gradient as it is — no LDS fill, no conversion, no read-modify-write.
    // Static segments: one count per pass-1 workgroup, read once here (the accumulation below reads them again from L2).
    bool any = n_dyn != 0u;
    if (!coarse && !any) {
      const uint32_t* cnts = buf.counts + (size_t)tile * G.segs;
      for (uint32_t sgi = threadIdx.x; sgi < G.segs; sgi += blockDim.x) any = any || cnts[sgi] != 0u;
    }
    if (!__syncthreads_or(any)) {
      if (threadIdx.x == 0) buf.dyn_cursor[tile] = 0u;
      return;
    }
  }
  {
    uint4* z = reinterpret_cast<uint4*>(acc);
    for (int e = threadIdx.x; e < entries; e += blockDim.x) z[e] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  PROBE_STAMP(0, 11);
  // self-cleaning cursor: the next call finds zeros again (the workspace state is zero-initialised once by its owner)
  if (threadIdx.x == 0) buf.dyn_cursor[tile] = 0u;
  const uint4* q = buf.queues + ((size_t)G.level_off[level] + (size_t)bin * Q);
  const int kk = fs.k;
  auto add_rec = [&](const uint4& r) {
    const float f0 = __uint_as_float(r.x), f1 = __uint_as_float(r.y);
    if (r.w & 0x80000000u) {  // x-pair: the x factor of the product ((g*wz)*wy)*wx is applied here
      const float wx = __uint_as_float(r.z), omx = 1.0f - wx;
      unsigned long long* a = acc + 2 * (r.w & 0x3fffu);
      unsigned long long* b = acc + 2 * ((r.w >> 14) & 0x3fffu);
      atomicAdd(a, to_fixed(f0 * omx, kk));  // ds_add_u64, no return
      atomicAdd(a + 1, to_fixed(f1 * omx, kk));
      atomicAdd(b, to_fixed(f0 * wx, kk));
      atomicAdd(b + 1, to_fixed(f1 * wx, kk));
    } else {
      unsigned long long* a = acc + 2 * (r.w & 0x3fffu);
      atomicAdd(a, to_fixed(f0, kk));
      atomicAdd(a + 1, to_fixed(f1, kk));
    }
  };
  if (!fs.empty && !fs.bad) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nw = blockDim.x >> 6;
    if (!coarse) {
      // static segments: wave w owns segments w, w + nw, ...; their counts are fetched up front (one load per lane),
      // so a trip of 4 segments has a single memory latency in front of its records, not two
      const uint32_t* cnts = buf.counts + (size_t)tile * G.segs;
      const uint32_t mine = G.segs > wave ? (G.segs - wave + nw - 1u) / nw : 0u;
      for (uint32_t c0 = 0; c0 < mine; c0 += 64u) {
        const uint32_t cv = (c0 + (uint32_t)lane < mine) ? cnts[wave + (c0 + (uint32_t)lane) * nw] : 0u;
        // Only the NON-EMPTY segments are visited (their lanes in a ballot; wave-uniform, scalar bit scans): an accumulating
        // call on few points (the proposal levels while few rays carry gradient) leaves most of a tile's 384 segments
        // empty, and walking them cost ~2.2 k clocks of pure instruction issue per trip of four
        // (scripts/probe_gated_scatter_clocks.py: 27 k clocks per tile with next to nothing to add).
        unsigned long long rem = __ballot(cv != 0u);
        // trips of 4 segments, software-pipelined in the source: the records of the next trip are requested before this one goes
        // through the LDS atomics. (The ISA does not keep them in flight — lane-predicated loads sit in branches, and the
        // compiler settles each trip with s_waitcnt vmcnt(0) —, but the unconditional form that does (clamped lanes re-reading
        // the segment's first record) measured 2-3 us SLOWER, as did the tile as two feature planes instead of (f0, f1) pairs:
        // with 16 waves per CU the latency is covered by the other waves, and what bounds the pass is the LDS atomic pipe plus
        // the VALU of the fixed-point conversions. profiles/r04_negative_results.txt.)
        uint32_t n[4], n_next[4], sg[4], sg_next[4];
        uint4 r[4][2], r_next[4][2];
        auto fetch = [&](uint32_t (&nn)[4], uint32_t (&ss)[4], uint4 (&rr)[4][2]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            nn[u] = 0u;
            ss[u] = 0u;
            if (rem != 0ull) {
              const int j = __builtin_ctzll(rem);
              rem &= rem - 1ull;
              nn[u] = (uint32_t)__builtin_amdgcn_readlane((int)cv, j);
              ss[u] = wave + (c0 + (uint32_t)j) * nw;
            }
            const uint4* seg = q + (size_t)ss[u] * C;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
              const uint32_t e = (uint32_t)lane + 64u * v;
              rr[u][v] = e < nn[u] ? rec_load(seg + e) : make_uint4(0u, 0u, 0u, 0u);
            }
          }
        };
        fetch(n, sg, r);
        while (n[0] != 0u) {  // (segments are taken in order: an empty first slot means none is left)
          fetch(n_next, sg_next, r_next);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int v = 0; v < 2; ++v)
              if ((uint32_t)lane + 64u * v < n[u]) add_rec(r[u][v]);
            const uint4* seg = q + (size_t)sg[u] * C;
            for (uint32_t e = (uint32_t)lane + 128u; e < n[u]; e += 64u) add_rec(rec_load(seg + e));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            n[u] = n_next[u];
            sg[u] = sg_next[u];
            r[u][0] = r_next[u][0];
            r[u][1] = r_next[u][1];
          }
        }
      }
    }
    PROBE_STAMP(0, 12);
    // dynamic area: contiguous, 4 records in flight per thread
    const uint4* dq = q + static_end;
    for (uint32_t e0 = 0; e0 < n_dyn; e0 += blockDim.x * 4u) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t e = e0 + (uint32_t)u * blockDim.x + threadIdx.x;
        r[u] = e < n_dyn ? rec_load(dq + e) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + (uint32_t)u * blockDim.x + threadIdx.x < n_dyn) add_rec(r[u]);
    }
    PROBE_STAMP(0, 13);
    // spill list (normally empty): every tile scans the folded prefix for its own records
    for (uint32_t e = threadIdx.x; e < n_spill; e += blockDim.x)
      if (buf.spill_tile[e] == tile) add_rec(buf.spill_rec[e]);
  }
  PROBE_STAMP(0, 14);
  __syncthreads();
  PROBE_STAMP(0, 15);
  float4* out = reinterpret_cast<float4*>(
      dtable + ((((size_t)level << grid.log2_table_size) + ((size_t)bin << G.slice_log2)) << 1));
  const uint4* a4 = reinterpret_cast<const uint4*>(acc);  // one entry = (lo0, hi0, lo1, hi1)
  const float nan = __uint_as_float(0x7fc00000u);
  for (int i = threadIdx.x; i < entries / 2; i += blockDim.x) {  // two entries = one float4 of the gradient
    const uint4 e0 = a4[2 * i], e1 = a4[2 * i + 1];
    float4 v;
    if (fs.bad) {
      v = make_float4(nan, nan, nan, nan);
    } else {
      v.x = from_fixed(((unsigned long long)e0.y << 32) | e0.x, fs.k);
      v.y = from_fixed(((unsigned long long)e0.w << 32) | e0.z, fs.k);
      v.z = from_fixed(((unsigned long long)e1.y << 32) | e1.x, fs.k);
      v.w = from_fixed(((unsigned long long)e1.w << 32) | e1.z, fs.k);
    }
    if (overwrite) {  // write-only gradient: no zero-fill before the call, no read here
      out[i] = v;
    } else if ((e0.x | e0.y | e0.z | e0.w | e1.x | e1.y | e1.z | e1.w) != 0u || fs.bad) {  // sole owner of the tile
      float4 o = out[i];
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      out[i] = o;
    }
  }
  PROBE_STAMP(0, 16);
}

template <bool kRider>
__global__ void scatter_apply_kernel(nsamd_grid grid, ScatterGeom G, ScatterBufs buf, float* __restrict__ dtable,
                                     int overwrite, const uint32_t* __restrict__ gate, ReduceRider rider) {
  scatter_apply_body<kRider>(grid, G, buf, dtable, overwrite, gate, rider);
}

struct ApplyCall {
  nsamd_grid grid;
  ScatterGeom G;
  ScatterBufs buf;
  float* dtable;
  int overwrite;
  const uint32_t* gate;
};

// two calls' pass 2 in one launch (see scatter_route_runs_pair_kernel); both with the same workgroup size and LDS tile
__global__ void scatter_apply_pair_kernel(ApplyCall a, ApplyCall b) {
  if (blockIdx.z == 0) {
    if (blockIdx.x >= (1u << a.G.log2_bins) || (int)blockIdx.y >= a.G.num_levels) return;
    scatter_apply_body<false>(a.grid, a.G, a.buf, a.dtable, a.overwrite, a.gate, ReduceRider{});
  } else {
    if (blockIdx.x >= (1u << b.G.log2_bins) || (int)blockIdx.y >= b.G.num_levels) return;
    scatter_apply_body<false>(b.grid, b.G, b.buf, b.dtable, b.overwrite, b.gate, ReduceRider{});
  }
}

// After pass 2: spill records beyond the folded prefix (a pathological batch) are applied with float atomics — exact
// sums, but in no fixed order, so they are counted; then the per-call header state goes back to zero.
__device__ __forceinline__ void scatter_finish_body(const nsamd_grid& grid, const ScatterGeom& G, const ScatterBufs& buf,
                                                    float* __restrict__ dtable, const uint32_t* __restrict__ gate) {
  if (gate_is_clear(gate)) return;  // nothing was routed: the per-call state is still zero
  const uint32_t total = buf.hdr[kHdrSpillCount];  // (may exceed the capacity: the excess went out directly)
  const uint32_t n = min(total, G.spill_cap);
  for (uint32_t e = kSpillFold + blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint4 r = buf.spill_rec[e];
    const uint32_t tile = buf.spill_tile[e];
    const uint32_t level = tile >> G.log2_bins, bin = tile & ((1u << G.log2_bins) - 1u);
    float* t = dtable + ((((size_t)level << grid.log2_table_size) + ((size_t)bin << G.slice_log2)) << 1);
    const float f0 = __uint_as_float(r.x), f1 = __uint_as_float(r.y);
    if (r.w & 0x80000000u) {
      const float wx = __uint_as_float(r.z), omx = 1.0f - wx;
      float* a = t + 2 * (size_t)(r.w & 0x3fffu);
      float* b = t + 2 * (size_t)((r.w >> 14) & 0x3fffu);
      unsafeAtomicAdd(a, f0 * omx);
      unsafeAtomicAdd(a + 1, f1 * omx);
      unsafeAtomicAdd(b, f0 * wx);
      unsafeAtomicAdd(b + 1, f1 * wx);
    } else {
      float* a = t + 2 * (size_t)(r.w & 0x3fffu);
      unsafeAtomicAdd(a, f0);
      unsafeAtomicAdd(a + 1, f1);
    }
  }
  __syncthreads();
  // the last workgroup to finish (every workgroup has read the counter by then) resets the per-call state
  if (threadIdx.x == 0 && atomicAdd(buf.hdr + kHdrTicket, 1u) == gridDim.x - 1) {
    for (int l = 0; l < NSAMD_MAX_LEVELS; ++l) buf.hdr[l] = 0u;
    if (total) {
      buf.hdr[kHdrEvtSpill] += total;
      if (n > kSpillFold) buf.hdr[kHdrEvtUnordered] += n - kSpillFold;
    }
    buf.hdr[kHdrSpillCount] = 0u;
    buf.hdr[kHdrTicket] = 0u;
  }
}

__global__ void scatter_finish_kernel(nsamd_grid grid, ScatterGeom G, ScatterBufs buf, float* __restrict__ dtable,
                                      const uint32_t* __restrict__ gate) {
  scatter_finish_body(grid, G, buf, dtable, gate);
}

__global__ void scatter_finish_pair_kernel(ApplyCall a, ApplyCall b) {  // (same grid for both calls)
  if (blockIdx.z == 0) scatter_finish_body(a.grid, a.G, a.buf, a.dtable, a.gate);
  else scatter_finish_body(b.grid, b.G, b.buf, b.dtable, b.gate);
}

// ---- host ---------------------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e != nullptr ? atoi(e) : dflt;
}
static unsigned apply_threads(int slice_log2);

// fine-kernel shape (threads, points per thread, levels per thread); NSAMD_SCATTER_SHAPE = "TPL" digits (experiments; read
// once). Measured on MI355X (profiles/r02_scatter_variants.txt): 1024 x 1 x 4 is the fastest or within noise of it; two
// points per thread, two levels per thread and an LDS-staged coalescing variant were slower and are gone.
struct FineShape {
  int threads, pts, levels;
};
static FineShape fine_shape() {
  static const int code = env_int("NSAMD_SCATTER_SHAPE", 114);
  switch (code) {
    case 514: return FineShape{512, 1, 4};
    case 112: return FineShape{1024, 1, 2};
    default: return FineShape{1024, 1, 4};
  }
}

static int scatter_target_tiles() {
  static const int v = env_int("NSAMD_SCATTER_TILES", 512);
  return v >= 64 ? v : 512;
}

constexpr int kSliceLog2Max = 13;  // 8192 entries x 2 x int64 = 128 KiB of the 160 KiB LDS

ScatterPlan scatter_plan(const nsamd_grid& grid, int64_t M, bool max_spill) {
  ScatterPlan p{};
  if (M <= 0 || grid.num_levels <= 0 || grid.num_levels > NSAMD_MAX_LEVELS) return p;
  int bits = 0;
  while ((grid.num_levels << bits) < scatter_target_tiles()) ++bits;
  int sl = grid.log2_table_size - bits;
  // a tile at least as wide as the finest resolution keeps every x-pair inside one tile (no straddling pairs)
  float res_max = 0.0f;
  for (int l = 0; l < grid.num_levels; ++l) res_max = grid.scalings[l] > res_max ? grid.scalings[l] : res_max;
  int sl_pair = 1;
  while ((1 << sl_pair) < (int)res_max + 2 && sl_pair < kSliceLog2Max) ++sl_pair;
  sl = sl < sl_pair ? sl_pair : sl;
  sl = sl > kSliceLog2Max ? kSliceLog2Max : (sl < 8 ? 8 : sl);
  if (sl > grid.log2_table_size) sl = grid.log2_table_size;
  const int log2_bins = grid.log2_table_size - sl;
  if (log2_bins > kMaxLog2Bins) return p;
  const FineShape fsx = fine_shape();
  ScatterGeom& g = p.geom;
  g.slice_log2 = sl;
  g.log2_bins = log2_bins;
  g.num_levels = grid.num_levels;
  g.block_points = (uint32_t)(fsx.threads * fsx.pts);
  const int64_t bins = (int64_t)1 << log2_bins;
  const int64_t segs = (M + g.block_points - 1) / g.block_points;
  // static segment: 2x the uniform-hash expectation of 4 pair records per point and level
  int64_t C = 2 * ((4 * (int64_t)g.block_points + bins - 1) / bins);
  C = (C + 3) & ~(int64_t)3;
  if (C < 16) C = 16;
  const int64_t expect = (4 * M + bins - 1) / bins;  // pair records per tile, uniform hash
  // Queue capacity per tile: static segments + a dynamic area, at least 2.5x the uniform-hash expectation. Levels whose
  // lattice (res + 1)^3 fills less than half of the table are SPARSE: their few reachable entries concentrate the updates
  // of whole regions of space on a handful of tiles (r02b: level 0 of the bench overflowed a 2.5x queue on every step),
  // so they get 8x.
  int64_t Qn = segs * C + expect / 2 + 64;
  if (Qn < 2 * expect + expect / 2 + 64) Qn = 2 * expect + expect / 2 + 64;  // run-mode levels use the whole queue
  Qn = (Qn + 3) & ~(int64_t)3;
  int64_t Qs = segs * C + 6 * expect + 64;
  if (Qs < 8 * expect + 64) Qs = 8 * expect + 64;
  Qs = (Qs + 3) & ~(int64_t)3;
  p.tiles = bins * grid.num_levels;
  int64_t total = 0, Qmax = 0;
  for (int l = 0; l < grid.num_levels; ++l) {
    const double lattice = ((double)grid.scalings[l] + 1.0) * ((double)grid.scalings[l] + 1.0) * ((double)grid.scalings[l] + 1.0);
    const int64_t Ql = lattice * 2.0 <= (double)((int64_t)1 << grid.log2_table_size) ? Qs : Qn;
    if (total >= 0x7fffffffLL) return p;
    g.level_off[l] = (uint32_t)total;
    g.level_cap[l] = (uint32_t)Ql;
    total += bins * Ql;
    Qmax = Ql > Qmax ? Ql : Qmax;
  }
  if (total >= 0x7fffffffLL || segs >= 0x7fffffffLL || C >= 0x3fffffffLL) return p;
  g.queue_records = (uint32_t)total;
  g.segs = (uint32_t)segs;
  g.seg_cap = (uint32_t)C;
  // spill list: worst case (every record of the call: 4 pairs, or after run merging at most as many singles, per point
  // and level) for write-only calls; otherwise a quarter of the expected total
  int64_t spill = 4 * M * grid.num_levels + 64;
  if (!max_spill) {
    const int64_t part = M * grid.num_levels + 4096;
    spill = spill < part ? spill : part;
  }
  if (spill >= 0x7fffffffLL) return p;
  g.spill_cap = (uint32_t)spill;
  if (Qmax + (int64_t)kSpillFold >= ((int64_t)1 << 30)) return p;  // the per-level headroom is derived in pass 2
  g.headroom = 0;
  g.coarse_mask = 0u;
  const int64_t cursor_words = (p.tiles + 3) & ~(int64_t)3;
  const int64_t count_words = (p.tiles * segs + 3) & ~(int64_t)3;
  p.state_words = kHdrWords + cursor_words;
  p.total_words = kHdrWords + cursor_words + count_words + 4 * total + 4 * spill + ((spill + 3) & ~(int64_t)3);
  p.ok = true;
  return p;
}

ScatterPlan scatter_plan_producers(const nsamd_grid& grid, int64_t M, int workgroups, int seg_cap) {
  ScatterPlan p{};
  if (M <= 0 || workgroups <= 0 || seg_cap < 16 || grid.num_levels <= 0 || grid.num_levels > NSAMD_MAX_LEVELS) return p;
  // tiles as scatter_plan chooses them: wide enough for every x-pair to stay inside one tile, 128 KiB of LDS at most
  int bits = 0;
  while ((grid.num_levels << bits) < scatter_target_tiles()) ++bits;
  int sl = grid.log2_table_size - bits;
  float res_max = 0.0f;
  for (int l = 0; l < grid.num_levels; ++l) res_max = grid.scalings[l] > res_max ? grid.scalings[l] : res_max;
  int sl_pair = 1;
  while ((1 << sl_pair) < (int)res_max + 2 && sl_pair < kSliceLog2Max) ++sl_pair;
  sl = sl < sl_pair ? sl_pair : sl;
  sl = sl > kSliceLog2Max ? kSliceLog2Max : (sl < 8 ? 8 : sl);
  if (sl > grid.log2_table_size) sl = grid.log2_table_size;
  const int log2_bins = grid.log2_table_size - sl;
  if (log2_bins > kProducerMaxLog2Bins) return p;  // the producer keeps one LDS counter per (level, tile)
  ScatterGeom& g = p.geom;
  g.slice_log2 = sl;
  g.log2_bins = log2_bins;
  g.num_levels = grid.num_levels;
  g.block_points = 0u;
  const int64_t bins = (int64_t)1 << log2_bins;
  const int64_t segs = workgroups, C = (seg_cap + 3) & ~3;
  const int64_t expect = (4 * M + bins - 1) / bins;  // pair records per tile, uniform hash
  // static segments (scripts/study_fused_route_overflow.py: a 256-record segment holds every level of the benchmark's
  // batches but ~60 records of level 0) + a dynamic area for what overflows them
  int64_t Q = segs * C + expect / 2 + 64;
  Q = (Q + 3) & ~(int64_t)3;
  p.tiles = bins * grid.num_levels;
  int64_t total = 0;
  for (int l = 0; l < grid.num_levels; ++l) {
    if (total >= 0x7fffffffLL) return p;
    g.level_off[l] = (uint32_t)total;
    g.level_cap[l] = (uint32_t)Q;
    total += bins * Q;
  }
  if (total >= 0x7fffffffLL || Q + (int64_t)kSpillFold >= ((int64_t)1 << 30)) return p;
  g.queue_records = (uint32_t)total;
  g.segs = (uint32_t)segs;
  g.seg_cap = (uint32_t)C;
  const int64_t spill = 4 * M * grid.num_levels + 64;  // worst case: the gradient is write-only, nothing may be lost
  if (spill >= 0x7fffffffLL) return p;
  g.spill_cap = (uint32_t)spill;
  g.headroom = 0;
  g.coarse_mask = 0u;
  const int64_t cursor_words = (p.tiles + 3) & ~(int64_t)3;
  const int64_t count_words = (p.tiles * segs + 3) & ~(int64_t)3;
  p.state_words = kHdrWords + cursor_words;
  p.total_words = kHdrWords + cursor_words + count_words + 4 * total + 4 * spill + ((spill + 3) & ~(int64_t)3);
  p.ok = true;
  return p;
}

// 128 KiB of dynamic LDS need the opt-in, per device
static int apply_lds_attribute() {
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NSAMD_ERR_NO_DEVICE;
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_apply_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16u << kSliceLog2Max)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_apply_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16u << kSliceLog2Max)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_apply_pair_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16u << kSliceLog2Max)) != hipSuccess)
      return NSAMD_ERR_LAUNCH;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  return NSAMD_OK;
}

int scatter_apply_launch(const nsamd_grid& grid, const ScatterPlan& plan, float* workspace, float* dtable, bool overwrite,
                         hipStream_t st, const ReduceRider* rider) {
  ScatterGeom G = plan.geom;
  ScatterBufs buf = scatter_bufs(workspace, plan);
  buf.log2_table_size = grid.log2_table_size;
  if (!overwrite) buf.direct_table = dtable;
  int rc = apply_lds_attribute();
  if (rc) return rc;
  const unsigned threads = apply_threads(G.slice_log2);
  ReduceRider rd{};
  unsigned extra_rows = 0u;
  if (rider != nullptr && rider->blocks > 0) {
    if (threads != (unsigned)kReduceThreads) return NSAMD_ERR_UNSUPPORTED;  // (callers check `scatter_apply_takes_rider` first)
    rd = *rider;
    extra_rows = ((unsigned)rd.blocks + (1u << G.log2_bins) - 1u) >> G.log2_bins;
  }
  dim3 g2(1u << G.log2_bins, (unsigned)grid.num_levels + extra_rows);
  if (extra_rows != 0u)
    scatter_apply_kernel<true><<<g2, threads, (size_t)16 << G.slice_log2, st>>>(grid, G, buf, dtable, overwrite ? 1 : 0, nullptr, rd);
  else
    scatter_apply_kernel<false><<<g2, threads, (size_t)16 << G.slice_log2, st>>>(grid, G, buf, dtable, overwrite ? 1 : 0, nullptr, rd);
  NSAMD_CHECK_LAUNCH();
  scatter_finish_kernel<<<32, 256, 0, st>>>(grid, G, buf, dtable, nullptr);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

// threads of an apply-pass workgroup for tiles of 2^slice_log2 entries. NSAMD_APPLY_THREADS_12 (experiments): the count for
// 4096-entry tiles (64 KiB of LDS: 512 threads let two workgroups share a compute unit, NSAMD_SCATTER_TILES=2048).
static unsigned apply_threads(int slice_log2) {
  static const int t12 = env_int("NSAMD_APPLY_THREADS_12", 1024);
  if (slice_log2 == 12 && (t12 == 256 || t12 == 512 || t12 == 1024)) return (unsigned)t12;
  return slice_log2 > 11 ? 1024u : (slice_log2 > 9 ? 512u : 256u);
}

bool scatter_apply_takes_rider(const ScatterPlan& plan) { return apply_threads(plan.geom.slice_log2) == 1024u; }

ScatterBufs scatter_bufs(float* workspace, const ScatterPlan& p) {
  ScatterBufs b;
  uint32_t* w = reinterpret_cast<uint32_t*>(workspace);
  b.hdr = w;
  b.dyn_cursor = w + kHdrWords;
  const int64_t cursor_words = (p.tiles + 3) & ~(int64_t)3;
  b.counts = b.dyn_cursor + cursor_words;
  const int64_t count_words = (p.tiles * (int64_t)p.geom.segs + 3) & ~(int64_t)3;
  b.queues = reinterpret_cast<uint4*>(b.counts + count_words);  // 16-B aligned: all sizes above are multiples of 4 words
  b.spill_rec = b.queues + (size_t)p.geom.queue_records;
  b.spill_tile = reinterpret_cast<uint32_t*>(b.spill_rec + p.geom.spill_cap);
  b.direct_table = nullptr;
  b.log2_table_size = 0;
  b.log2_bins = p.geom.log2_bins;
  b.slice_log2 = p.geom.slice_log2;
  return b;
}

template <int kThreads, int kPts, int kLevels>
static void launch_fine(const nsamd_points& pts, int64_t M, int transform, const nsamd_aabb& aabb, const nsamd_grid& grid,
                        const float* denc, int64_t stride_p, int64_t stride_k, const ScatterGeom& G,
                        const LevelList& fine, const ScatterBufs& buf, const uint32_t* gate, const uint8_t* ray_mask,
                        hipStream_t st) {
  const size_t lds = sizeof(uint32_t) * (3 * (size_t)kLevels * ((size_t)1 << G.log2_bins) + kLevels);
  dim3 g1(G.segs, (unsigned)((fine.count + kLevels - 1) / kLevels));
  scatter_route_fine_kernel<kThreads, kPts, kLevels><<<g1, kThreads, lds, st>>>(pts, M, transform, aabb, grid, denc,
                                                                                 stride_p, stride_k, G, fine, buf, gate, ray_mask);
}

// Which levels of a call go through the run-merging kernel (coarse) and which through the plain route (fine); sets G.coarse_mask.
static void classify_levels(const nsamd_points& pts, const nsamd_grid& grid, ScatterGeom& G, LevelList& coarse,
                            LevelList& fine) {
  static const int combine_env = env_int("NSAMD_SCATTER_COMBINE_RES", 0);
  float coarse_below = 0.0f;
  // Round 6 (profiles/r06_s12_*, r06_s14_*): since the run kernel and the apply pass on non-empty segments (rounds 3 - 5) the
  // 96-sample level is cheaper merged as well — its route + apply read 40 + 37 us plain against 35 + 9 us for the 2.7 x larger
  // 256-sample level merged; all five of its levels through the run kernel: long run 0.6919 / 0.6883 against 0.6971 / 0.6959 ms
  // from step 40, window 0.657 / 0.658 against 0.667 / 0.667 from step 0 (NSAMD_SCATTER_MERGE_96=0: the plain route, A/B).
  static const int merge96 = env_int("NSAMD_SCATTER_MERGE_96", 1);
  if (pts.positions == nullptr && pts.samples_per_ray >= (merge96 ? 96 : 192))
    coarse_below = pts.samples_per_ray >= 192 ? (float)pts.samples_per_ray : 1e30f;
  if (combine_env > 0) coarse_below = (float)combine_env;
  for (int l = 0; l < grid.num_levels; ++l) {
    if (grid.scalings[l] < coarse_below) {
      G.coarse_mask |= 1u << l;
      coarse.level[coarse.count++] = (int8_t)l;
    } else {
      fine.level[fine.count++] = (int8_t)l;
    }
  }
}

// levels per thread of the run kernel (NSAMD_RUNS_LEVELS = 1 / 2 / 4, read once): fewer levels per thread = more, shorter
// workgroups — its time is the latency of one workgroup's two sweeps (profiles/r03_sparse_regime_kernel_stats.csv)
// (MI355X, driver window of the bench: 4 -> 75 us, 2 -> 54, 1 -> 54 per launch of the 256-sample level's scatter)
static int runs_levels_setting() {
  static const int runs_levels = env_int("NSAMD_RUNS_LEVELS", 2);
  return runs_levels == 1 || runs_levels == 2 ? runs_levels : 4;
}

int scatter_launch(const nsamd_points& pts, int64_t M, int transform, const nsamd_aabb& aabb, const nsamd_grid& grid,
                   const float* denc, int64_t stride_p, int64_t stride_k, float* dtable, float* workspace,
                   const ScatterPlan& plan, bool overwrite, const uint32_t* gate, const uint8_t* ray_mask, hipStream_t st) {
  if (gate != nullptr && overwrite) return NSAMD_ERR_INVALID_ARG;  // a write-only gradient must always be written
  if (ray_mask != nullptr && (gate == nullptr || pts.positions != nullptr)) return NSAMD_ERR_INVALID_ARG;  // ray mode only
  ScatterGeom G = plan.geom;
  ScatterBufs buf = scatter_bufs(workspace, plan);
  buf.log2_table_size = grid.log2_table_size;
  if (!overwrite) buf.direct_table = dtable;
  // Levels whose cells are wide against the sample spacing go through the run-merging kernel: with >= 192 samples per
  // ray (the first proposal level) consecutive samples share cells on every level of the small proposal grids and
  // merging pays (75 vs 107 us at M = 1 M); with 96 or 48 samples per ray the plain route is faster on every level
  // (65 vs 88 us, 178 vs 197 us for the main table; profiles/r02a_*). NSAMD_SCATTER_COMBINE_RES > 0 overrides the
  // threshold: levels with resolution below it are merged (1 = none).
  LevelList coarse{}, fine{};
  classify_levels(pts, grid, G, coarse, fine);
  {
    const int rc = apply_lds_attribute();
    if (rc) return rc;
  }
  if (fine.count > 0) {
    const FineShape s = fine_shape();
    const int code = s.threads * 100 + s.pts * 10 + s.levels;
    switch (code) {
      case 51214: launch_fine<512, 1, 4>(pts, M, transform, aabb, grid, denc, stride_p, stride_k, G, fine, buf, gate, ray_mask, st); break;
      case 102412: launch_fine<1024, 1, 2>(pts, M, transform, aabb, grid, denc, stride_p, stride_k, G, fine, buf, gate, ray_mask, st); break;
      default: launch_fine<1024, 1, 4>(pts, M, transform, aabb, grid, denc, stride_p, stride_k, G, fine, buf, gate, ray_mask, st); break;
    }
    NSAMD_CHECK_LAUNCH();
  }
  if (coarse.count > 0) {
    const int runs_levels = runs_levels_setting();
    const int64_t per_block = (int64_t)kRunThreads * kRunLen;
    auto launch_runs = [&](auto tag) {
      constexpr int kL = decltype(tag)::value;
      const size_t lds = sizeof(uint32_t) * (3 * (size_t)kL * ((size_t)1 << G.log2_bins) + kL);
      dim3 g1((unsigned)((M + per_block - 1) / per_block), (unsigned)((coarse.count + kL - 1) / kL));
      scatter_route_runs_kernel<kL><<<g1, kRunThreads, lds, st>>>(pts, M, transform, aabb, grid, denc, stride_p, stride_k, G,
                                                                coarse, buf, gate, ray_mask);
    };
    if (runs_levels == 1) launch_runs(std::integral_constant<int, 1>{});
    else if (runs_levels == 2) launch_runs(std::integral_constant<int, 2>{});
    else launch_runs(std::integral_constant<int, 4>{});
    NSAMD_CHECK_LAUNCH();
  }
  const unsigned threads = apply_threads(G.slice_log2);
  dim3 g2(1u << G.log2_bins, (unsigned)grid.num_levels);
  scatter_apply_kernel<false><<<g2, threads, (size_t)16 << G.slice_log2, st>>>(grid, G, buf, dtable, overwrite ? 1 : 0, gate,
                                                                               ReduceRider{});
  NSAMD_CHECK_LAUNCH();
  scatter_finish_kernel<<<32, 256, 0, st>>>(grid, G, buf, dtable, gate);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

int scatter_launch_pair(const ScatterCall& a, const ScatterCall& b, hipStream_t st) {
  const ScatterCall* c[2] = {&a, &b};
  ScatterGeom G[2];
  ScatterBufs buf[2];
  LevelList coarse[2], fine[2];
  for (int i = 0; i < 2; ++i) {
    if (c[i]->gate != nullptr && c[i]->overwrite) return NSAMD_ERR_INVALID_ARG;
    if (c[i]->ray_mask != nullptr && (c[i]->gate == nullptr || c[i]->pts.positions != nullptr)) return NSAMD_ERR_INVALID_ARG;
    if (!c[i]->plan.ok || c[i]->M <= 0) return NSAMD_ERR_UNSUPPORTED;
    G[i] = c[i]->plan.geom;
    buf[i] = scatter_bufs(c[i]->workspace, c[i]->plan);
    buf[i].log2_table_size = c[i]->grid.log2_table_size;
    if (!c[i]->overwrite) buf[i].direct_table = c[i]->dtable;
    coarse[i] = LevelList{};
    fine[i] = LevelList{};
    classify_levels(c[i]->pts, c[i]->grid, G[i], coarse[i], fine[i]);
    // only calls that route every level through the run kernel are merged (the proposal levels of nerfacto: 256 and 96
    // samples per ray on small grids); anything else goes through scatter_launch, call by call
    if (fine[i].count != 0 || coarse[i].count == 0) return NSAMD_ERR_UNSUPPORTED;
  }
  if (a.workspace == b.workspace || a.dtable == b.dtable) return NSAMD_ERR_UNSUPPORTED;  // shared state: one after the other
  const unsigned threads = apply_threads(G[0].slice_log2);
  if (threads != apply_threads(G[1].slice_log2)) return NSAMD_ERR_UNSUPPORTED;
  {
    const int rc = apply_lds_attribute();
    if (rc) return rc;
  }
  const int64_t per_block = (int64_t)kRunThreads * kRunLen;
  auto launch_runs = [&](auto tag) {
    constexpr int kL = decltype(tag)::value;
    RunsCall rc[2];
    size_t lds = 0;
    unsigned gx = 0, gy = 0;
    for (int i = 0; i < 2; ++i) {
      rc[i] = RunsCall{c[i]->pts, c[i]->M, c[i]->transform, c[i]->aabb, c[i]->grid, c[i]->denc, c[i]->stride_p, c[i]->stride_k,
                       G[i], coarse[i], buf[i], c[i]->gate, c[i]->ray_mask,
                       (uint32_t)((c[i]->M + per_block - 1) / per_block), (uint32_t)((coarse[i].count + kL - 1) / kL)};
      const size_t l = sizeof(uint32_t) * (3 * (size_t)kL * ((size_t)1 << G[i].log2_bins) + kL);
      lds = l > lds ? l : lds;
      gx = rc[i].grid_x > gx ? rc[i].grid_x : gx;
      gy = rc[i].grid_y > gy ? rc[i].grid_y : gy;
    }
    scatter_route_runs_pair_kernel<kL><<<dim3(gx, gy, 2u), kRunThreads, lds, st>>>(rc[0], rc[1]);
  };
  const int runs_levels = runs_levels_setting();
  if (runs_levels == 1) launch_runs(std::integral_constant<int, 1>{});
  else if (runs_levels == 2) launch_runs(std::integral_constant<int, 2>{});
  else launch_runs(std::integral_constant<int, 4>{});
  NSAMD_CHECK_LAUNCH();
  ApplyCall ac[2];
  unsigned bins = 0, levels = 0;
  size_t lds = 0;
  for (int i = 0; i < 2; ++i) {
    ac[i] = ApplyCall{c[i]->grid, G[i], buf[i], c[i]->dtable, c[i]->overwrite ? 1 : 0, c[i]->gate};
    bins = (1u << G[i].log2_bins) > bins ? (1u << G[i].log2_bins) : bins;
    levels = (unsigned)c[i]->grid.num_levels > levels ? (unsigned)c[i]->grid.num_levels : levels;
    const size_t l = (size_t)16 << G[i].slice_log2;
    lds = l > lds ? l : lds;
  }
  scatter_apply_pair_kernel<<<dim3(bins, levels, 2u), threads, lds, st>>>(ac[0], ac[1]);
  NSAMD_CHECK_LAUNCH();
  scatter_finish_pair_kernel<<<dim3(32u, 1u, 2u), 256, 0, st>>>(ac[0], ac[1]);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

}  // namespace nsamd
