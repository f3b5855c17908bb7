// Internal interface of the table-gradient scatter (scatter.hip), used by hashgrid.hip's backward entry points.
// Not part of the C ABI.
#pragma once

#include "common.h"

namespace nsamd {

// ---- order-independent accumulation -------------------------------------------------------------------------------
// Float sums depend on their order, and every parallel scatter order is a race; Adam (eps = 1e-15) then turns the
// rounding residue of a cancelling sum into a full-size step, so two identical training runs drift apart within tens
// of steps. The scatter therefore accumulates in 64-bit FIXED POINT: integer addition is
// associative, so the result is bit-identical whatever the order of the atomics, the queue layout or which path
// (static segment / dynamic area / spill list) a record took. Scale per hash level: with `max` the largest |value| a
// record of the level can carry (exponent field e_max) and `headroom` bits for the number of summands,
//   fixed(v) = trunc(v * 2^k),  k = 188 - headroom - e_max   =>   |v * 2^k| < 2^(62 - headroom), |sum| < 2^62.
// A value keeps its full 24-bit mantissa while it is larger than max * 2^-(38 - headroom); smaller ones lose low bits
// gradually (absolute error <= 2^-k per summand, i.e. < 1e-12 of the level's largest gradient for headroom 20).
struct FixedScale {
  int k;       // value = fixed * 2^-k
  bool empty;  // nothing recorded on this level
  bool bad;    // non-finite gradient on this level: the output is NaN
};

__device__ __forceinline__ FixedScale fixed_scale(uint32_t max_bits, int headroom) {
  FixedScale s;
  const int e_max = (int)((max_bits >> 23) & 0xffu);
  s.empty = e_max == 0;  // zero (or denormal: < 1.2e-38, flushed) everywhere
  s.bad = e_max == 255;
  s.k = 188 - headroom - e_max;
  return s;
}

// trunc(v * 2^k) as a 64-bit two's-complement integer, |v * 2^k| < 2^44 (headroom >= 18). 8 VALU operations: the scaled
// value is split into a high part (multiple of 2^24) and the rest, both exactly representable, each converted with the
// hardware float -> int32 conversion (round toward zero: symmetric, -0 -> 0) and rejoined by one 64-bit multiply-add.
#ifndef NSAMD_FIXED_ROUND
#define NSAMD_FIXED_ROUND 0
#endif
__device__ __forceinline__ unsigned long long to_fixed(float v, int k) {
  const float t = ldexpf(v, k);                          // exact (or flushed to zero when denormal)
  const float hi_f = truncf(t * 5.9604644775390625e-8f); // t * 2^-24
  const float lo_f = fmaf(hi_f, -16777216.0f, t);        // exact: t minus its high part
#if NSAMD_FIXED_ROUND
  // round-to-nearest-even of the low part (one v_rndne_f32 more): |error| <= 2^-(k+1) per summand and unbiased, where the
  // truncation's error is one-sided (toward zero, <= 2^-k). Measured against each other on the PSNR stand-in
  // (profiles/r05_psnr_ab.txt); a build-time switch because the sums' bits differ.
  const long long r = (long long)(int)hi_f * 16777216ll + (long long)(int)rintf(lo_f);
#else
  const long long r = (long long)(int)hi_f * 16777216ll + (long long)(int)lo_f;
#endif
  return (unsigned long long)r;  // the atomics add modulo 2^64
}

__device__ __forceinline__ float from_fixed(unsigned long long a, int k) {
  return (float)ldexp((double)(long long)a, -k);
}

// ---- geometry -----------------------------------------------------------------------------------------------------
struct LevelList {
  int8_t level[32];
  int count;
};

// The table gradient is partitioned into (level, tile) pieces of 2^slice_log2 entries; a tile's queue holds `tile_cap`
// 16-B records: first `segs * seg_cap` slots in STATIC segments (one per pass-1 workgroup: no reservation atomics),
// then a dynamic area handed out by a per-tile cursor (levels routed in run mode use the whole queue dynamically).
struct ScatterGeom {
  int32_t slice_log2, log2_bins, num_levels, headroom;
  uint32_t segs;         // pass-1 workgroups along the points = ceil(M / block_points)
  uint32_t block_points; // points per pass-1 workgroup of the fine kernel
  uint32_t seg_cap;      // records per static segment
  uint32_t level_cap[NSAMD_MAX_LEVELS];  // records per tile queue of a level (sparse coarse levels get more: hot tiles)
  uint32_t level_off[NSAMD_MAX_LEVELS];  // first record of the level's queues: queue(level, bin) = off + bin * cap
  uint32_t queue_records;                // total
  uint32_t spill_cap;    // records of the spill list
  uint32_t coarse_mask;  // bit l: level l is routed by the run kernel (no static segments)
};

struct ScatterBufs {
  uint32_t* hdr;         // [kHdrWords]
  uint32_t* dyn_cursor;  // [tiles]
  uint32_t* counts;      // [tiles][segs]
  uint4* queues;         // [tiles][tile_cap]
  uint4* spill_rec;      // [spill_cap]
  uint32_t* spill_tile;  // [spill_cap]
  // last resort of an ACCUMULATING call whose (bounded) spill list is full: float atomics straight into the gradient
  // (nullptr for write-only calls, whose list holds the worst case)
  float* direct_table;
  int32_t log2_table_size, log2_bins, slice_log2;
};

constexpr int kHdrWords = 64;
constexpr int kHdrSpillCount = 32;   // records appended to the spill list by this call
constexpr int kHdrTicket = 33;       // finish kernel: last workgroup resets the header
constexpr int kHdrEvtSpill = 40;     // sticky: spill records seen since the workspace was created
constexpr int kHdrEvtUnordered = 41; // sticky: spill records applied with float atomics (beyond the fold limit)
constexpr int kHdrEvtLost = 42;      // sticky: records that found no room at all (cannot happen with a worst-case list)
constexpr int kProducerMaxLog2Bins = 6;  // producer kernels keep [levels][2^log2_bins] rank counters in LDS (4 KiB)
constexpr uint32_t kSpillFold = 8192;  // spill records pass 2 folds into its tiles (exact, order-independent)

#if defined(__HIPCC__)
// ---- device helpers shared by the route kernels (scatter.hip) and the field backward's record emission (field_mlp.hip) ----
// Queue records are written once (pass 1) and read once (pass 2, another launch): streaming accesses, kept out of the way of
// the table rows and gradients that do get re-used (NSAMD_SCATTER_NT=0 at build time: plain accesses, for A/B).
#ifndef NSAMD_SCATTER_NT
#define NSAMD_SCATTER_NT 1
#endif
typedef uint32_t rec_vec __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rec_store(uint4* dst, const uint4& r) {
  // (nontemporal STORES: 175 -> 399 us for the main table — the scattered 16-B records lose L2's write combining)
  // a GLOBAL store whatever the pointer's provenance (common.h, global_ptr): a producer that keeps the queue pointer in LDS
  // would otherwise issue flat stores
  rec_vec v;
  v.x = r.x, v.y = r.y, v.z = r.z, v.w = r.w;
  *global_ptr(reinterpret_cast<rec_vec*>(dst)) = v;
}
__device__ __forceinline__ uint4 rec_load(const uint4* src) {
#if NSAMD_SCATTER_NT
  const rec_vec v = __builtin_nontemporal_load(global_ptr(reinterpret_cast<const rec_vec*>(src)));
#else
  const rec_vec v = *global_ptr(reinterpret_cast<const rec_vec*>(src));
#endif
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Append a record that found no room in its tile (or an x-pair straddling two tiles). One returning atomic per
// wavefront; out of line: this is the cold path of ~50 call sites. False when the list is full.
static __device__ __noinline__ bool spill_list_append(uint32_t* hdr, uint4* spill_rec, uint32_t* spill_tile, uint32_t cap,
                                               uint32_t tile, uint4 rec) {
  const unsigned long long active = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(active);
  const uint32_t n = (uint32_t)__builtin_popcountll(active);
  const uint32_t mine = (uint32_t)__builtin_popcountll(active & ((1ull << lane) - 1ull));
  uint32_t base = 0u;
  if (lane == leader) base = atomicAdd(hdr + kHdrSpillCount, n);
  base = __shfl(base, leader);
  const uint32_t pos = base + mine;
  if (pos >= cap) return false;
  spill_rec[pos] = rec;
  spill_tile[pos] = tile;
  return true;
}

// Last resort of an ACCUMULATING call whose (bounded) spill list is full: float atomics straight into the gradient —
// exact, but in no fixed order (counted). `table_level_tile` = start of the tile in the gradient.
static __device__ __noinline__ void spill_direct(float* t, uint32_t* hdr, uint4 rec) {
  const float f0 = __uint_as_float(rec.x), f1 = __uint_as_float(rec.y);
  if (rec.w & 0x80000000u) {
    const float wx = __uint_as_float(rec.z), omx = 1.0f - wx;
    float* a = t + 2 * (size_t)(rec.w & 0x3fffu);
    float* b = t + 2 * (size_t)((rec.w >> 14) & 0x3fffu);
    unsafeAtomicAdd(a, f0 * omx);
    unsafeAtomicAdd(a + 1, f1 * omx);
    unsafeAtomicAdd(b, f0 * wx);
    unsafeAtomicAdd(b + 1, f1 * wx);
  } else {
    float* a = t + 2 * (size_t)(rec.w & 0x3fffu);
    unsafeAtomicAdd(a, f0);
    unsafeAtomicAdd(a + 1, f1);
  }
  atomicAdd(hdr + kHdrEvtUnordered, 1u);
}

__device__ __forceinline__ void spill_append(const ScatterBufs& buf, uint32_t cap, uint32_t tile, const uint4& rec) {
  if (spill_list_append(buf.hdr, buf.spill_rec, buf.spill_tile, cap, tile, rec)) return;
  if (buf.direct_table != nullptr) {
    const uint32_t level = tile >> buf.log2_bins, bin = tile & ((1u << buf.log2_bins) - 1u);
    spill_direct(buf.direct_table + ((((size_t)level << buf.log2_table_size) + ((size_t)bin << buf.slice_log2)) << 1),
                 buf.hdr, rec);
  } else {
    atomicAdd(buf.hdr + kHdrEvtLost, 1u);  // cannot happen: write-only calls size the list for the worst case
  }
}

struct PairHash {
  uint32_t ia, ib;
};

// hashes of the x-pair q (bit0: y is ceil, bit1: z is ceil) of a cell
__device__ __forceinline__ PairHash pair_hash(const Cell& c, int q, uint32_t mask) {
  const uint32_t yz = ((uint32_t)((q & 1) ? c.hi[1] : c.lo[1]) * kPrimeY) ^ ((uint32_t)((q & 2) ? c.hi[2] : c.lo[2]) * kPrimeZ);
  return PairHash{((uint32_t)c.lo[0] ^ yz) & mask, ((uint32_t)c.hi[0] ^ yz) & mask};
}

#endif  // __HIPCC__

struct ScatterPlan {
  bool ok;
  ScatterGeom geom;
  int64_t tiles;
  int64_t state_words;  // leading words that must be zero before the first call (header + cursors)
  int64_t total_words;
};

// Host: geometry for (grid, M); `max_spill` = true sizes the spill list for the worst case (write-only calls).
ScatterPlan scatter_plan(const nsamd_grid& grid, int64_t M, bool max_spill);

// Host: geometry for records emitted by `workgroups` PERSISTENT producer workgroups (the main field's backward kernel emits
// the pass-1 records of its own points: nsamd_field_mlp_bwd_scatter): one static segment of `seg_cap` records per
// (tile, workgroup) — a workgroup's LDS rank IS the slot —, a dynamic area behind them, a worst-case spill list.
ScatterPlan scatter_plan_producers(const nsamd_grid& grid, int64_t M, int workgroups, int seg_cap);

// Host: the workspace's regions for a plan.
ScatterBufs scatter_bufs(float* workspace, const ScatterPlan& p);

// Host: enqueue apply (pass 2) + finish for records that are already in the queues of `plan` (every level routed through
// static segments + dynamic area; counts[tile][seg] and hdr[level] written by the producer).
// `rider` (nullable): the main field's weight-gradient reduce carried along as extra workgroups of the apply pass
// (field_reduce.h); only where `scatter_apply_takes_rider(plan)`.
struct ReduceRider;
bool scatter_apply_takes_rider(const ScatterPlan& plan);
int scatter_apply_launch(const nsamd_grid& grid, const ScatterPlan& plan, float* workspace, float* dtable, bool overwrite,
                         hipStream_t stream, const ReduceRider* rider = nullptr);

// Host: enqueue route (pass 1) + apply (pass 2) + finish on `stream`. Returns an nsamd_status. `gate` (nullable, device):
// accumulating calls only — while *gate == 0 all kernels return at once (the gradient being scattered is all zeros);
// `ray_mask` (nullable, [rays] bytes, ray mode + gate only): samples of rays whose byte is 0 are zeros and are not loaded.
int scatter_launch(const nsamd_points& pts, int64_t M, int transform, const nsamd_aabb& aabb, const nsamd_grid& grid,
                   const float* denc, int64_t stride_p, int64_t stride_k, float* dtable, float* workspace,
                   const ScatterPlan& plan, bool overwrite, const uint32_t* gate, const uint8_t* ray_mask,
                   hipStream_t stream);

// Host: TWO independent calls (different tables, workspaces and gradients) with their route, apply and finish passes merged
// pairwise into one launch each. NSAMD_ERR_UNSUPPORTED (nothing enqueued): the calls cannot be merged — the caller issues
// scatter_launch for each.
struct ScatterCall {
  nsamd_points pts;
  int64_t M;
  int transform;
  nsamd_aabb aabb;
  nsamd_grid grid;
  const float* denc;
  int64_t stride_p, stride_k;
  float* dtable;
  float* workspace;
  ScatterPlan plan;
  bool overwrite;
  const uint32_t* gate;
  const uint8_t* ray_mask;
};
int scatter_launch_pair(const ScatterCall& a, const ScatterCall& b, hipStream_t stream);

}  // namespace nsamd
