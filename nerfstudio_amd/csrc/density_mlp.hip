// Proposal-network density head for gfx950: MLP IN -> H (ReLU) -> 1, density = avg * trunc_exp(.) * selector.
// Reference: HashMLPDensityField.get_density, /root/reference/nerfstudio/fields/density_fields.py:104-117;
// MLP.pytorch_fwd field_components/mlp.py:160-179; trunc_exp field_components/activations.py:28-54.
//
// 176 MACs per point (IN=10, H=16) is far too little for MFMA tiles to pay (the 16x16x4 tile would be 90 % padding
// on the output layer); the f32 vector rate equals the f32 MFMA rate on gfx950 anyway. So: one point per lane,
// feature-major enc[k][p] makes every load a coalesced 256-B row, the weights are wave-uniform and arrive through
// the scalar cache (s_load -> v_fmac with an SGPR operand), no LDS in the forward.
// Backward: the per-point input gradient is the same shape of work; the weight gradients need a sum over points,
// done per 256-point chunk through LDS ([feature][point] rows, stride 257). dW0 = Gh X^T (H x IN, k = points) IS a
// GEMM with a long k axis: each wave takes 64 of the chunk's points through 16 v_mfma_f32_16x16x4_f32 per row tile
// (operands are the LDS rows as they stand) and keeps its partial in registers across the chunks of a persistent
// workgroup — the first version gave each of 160 threads a 256-long dot product (1536 LDS read instructions per chunk
// against 128 now). One LDS reduction over the 4 waves at the end; the workgroup's partial goes to a scratch row that
// density_dw_reduce_kernel sums in a fixed order (bit-reproducible; without scratch: <= kMaxBlocks float atomics per
// weight element).
#include "common.h"
#include "density_point.h"
#include "proposal_chain.h"

namespace nsamd {

constexpr int kMlpBlock = 256;
constexpr int kMaxBlocks = 768;      // three 44-KB workgroups per compute unit
constexpr int kDensityActMax = 2048;  // bytes of the per-workgroup chunk-activity table (density_mlp_bwd_kernel)

// floats per workgroup row of the weight-gradient partial buffer: [dW0 | db0 | dW1 | db1], padded to 16 B
__host__ __device__ constexpr int density_partial_stride(int in_dim, int hidden) {
  return (hidden * in_dim + 2 * hidden + 1 + 3) & ~3;
}

template <int IN, int H>
__global__ __launch_bounds__(kMlpBlock) void density_mlp_fwd_kernel(const float* __restrict__ enc,
                                                                    const float* __restrict__ selector, int64_t M,
                                                                    nsamd_density_mlp mlp,
                                                                    float* __restrict__ density,
                                                                    float* __restrict__ pre_out) {
  const float* __restrict__ W0 = mlp.W0;
  const float* __restrict__ b0 = mlp.b0;
  const float* __restrict__ W1 = mlp.W1;
  for (int64_t p = (int64_t)blockIdx.x * kMlpBlock + threadIdx.x; p < M; p += (int64_t)gridDim.x * kMlpBlock) {
    float x[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = enc[(int64_t)k * M + p];
    float out = mlp.b1[0];
    constexpr int JU = (H <= 16) ? H : 4;
#pragma unroll JU
    for (int j = 0; j < H; ++j) {
      float a = b0[j];
#pragma unroll
      for (int k = 0; k < IN; ++k) a = fmaf(W0[j * IN + k], x[k], a);
      out = fmaf(W1[j], fmaxf(a, 0.0f), out);
    }
    const float sel = selector ? selector[p] : 1.0f;
    if (pre_out) pre_out[p] = out;
    density[p] = mlp.average_init_density * expf(out) * sel;
  }
}

// Fused proposal-network forward: contraction -> hash grid (all levels) -> MLP -> trunc_exp in ONE kernel, one point per
// lane (HashMLPDensityField.get_density, fields/density_fields.py:94-117). The encoded features stay in registers: the
// 4 * IN bytes per point of the feature-major `enc` buffer are neither written nor read back on the steps where the
// proposal networks get no gradient (ray_samplers.py:590: most steps after warm-up), and only written on the others
// (the backward's weight gradient needs them). Same operation order as hash_encode_fwd_kernel + density_mlp_fwd_kernel:
// bit-identical outputs. The proposal tables (5 levels x 2^17 entries x 8 B = 5 MB) sit in every XCD's L2.
// (The lane-pair arrangement that takes the main grid's forward from 76 to 63 us is SLOWER here — 37.2 -> 46.6 us on the 256-sample
// level, profiles/r05_s16_*: the proposal grids are coarse against the sample spacing and adjacent lanes already share their
// lines; csrc/experiments/rounds2to5_opt_in_variants.patch.)
template <int LEVELS, int H>
__global__ __launch_bounds__(kMlpBlock) void density_field_fwd_kernel(nsamd_points P, int64_t M, int transform,
                                                                      nsamd_aabb box, const float2* __restrict__ table,
                                                                      nsamd_grid grid, nsamd_density_mlp mlp,
                                                                      float* __restrict__ enc_out,
                                                                      float* __restrict__ selector_out,
                                                                      float* __restrict__ density,
                                                                      float* __restrict__ pre_out) {
  const int64_t p = (int64_t)blockIdx.x * kMlpBlock + threadIdx.x;
  if (p >= M) return;
  float x, y, z;
  load_position(P, p, x, y, z);
  density_point<LEVELS, H>(x, y, z, p, M, transform, box, table, grid, mlp, enc_out, selector_out, density, pre_out);
}

// (a device body: launched by density_mlp_bwd_kernel and, two independent calls side by side, by density_mlp_bwd_pair_kernel;
//  `nblocks` = the workgroups of THIS call, blockIdx.x < nblocks)
template <int IN, int H>
__device__ __forceinline__ void density_mlp_bwd_body(
    const float* __restrict__ enc, const float* __restrict__ selector, const float* __restrict__ pre,
    const float* __restrict__ ddensity, int64_t M, const nsamd_density_mlp& mlp, float* __restrict__ denc,
    float* __restrict__ dW0, float* __restrict__ db0, float* __restrict__ dW1, float* __restrict__ db1,
    float* __restrict__ partials, const uint32_t* __restrict__ gate, const uint8_t* __restrict__ ray_mask, int spr,
    int nblocks) {
  // Gated call (nsamd_density_mlp_bwd_gated): the flag nsamd_weights_bwd_gate raises when any ray of the level carries
  // gradient is clear -> every upstream gradient is an exact zero, so are all results of this launch; the zero-filled
  // weight gradients stay as they are and nothing downstream (gated the same way) reads `denc`.
  if (gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  constexpr int LD = kMlpBlock + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* gh_T = lds;            // [H][LD]   dL/d(hidden pre-activation) per point
  float* x_T = lds + H * LD;    // [IN][LD]  encoded features per point
  float* hg_T = x_T + IN * LD;  // [H][LD]   relu(hidden) * dL/dpre  (for dW1)
  float* gp = hg_T + H * LD;    // [LD]      dL/dpre per point
  // Weights: 193 wave-uniform scalars do not fit the SGPR file (the compiler spilled 147 of them to VGPR lanes);
  // staged once in LDS as rows of INP = IN rounded up to 4 floats and read back as broadcast ds_read_b128.
  constexpr int INP = (IN + 3) & ~3;
  float* w0s = lds + (((2 * H + IN + 1) * LD + 3) & ~3);  // [H][INP], 16-B aligned
  float* b0s = w0s + H * INP;   // [H]
  float* w1s = b0s + H;         // [H]
  // A workgroup owns a CONTIGUOUS range of 256-point chunks, so the rays it can touch are one contiguous range of the per-ray
  // mask: ONE burst of byte loads decides, before anything else is fetched, which of its chunks carry gradient. A workgroup
  // without any (the usual case while the interlevel loss reaches few rays) publishes a zero row of partial sums and leaves —
  // one memory round trip; it used to stage the weights (three round trips) and then look its chunks' rays up one load at a
  // time (22.4 -> 20.5 us per sparse launch of the 256-sample level).
  const int64_t chunks = (M + kMlpBlock - 1) / kMlpBlock;
  const int64_t per_wg = (chunks + nblocks - 1) / nblocks;
  const int64_t c_lo = (int64_t)blockIdx.x * per_wg;
  const int64_t c_hi = c_lo + per_wg < chunks ? c_lo + per_wg : chunks;
  constexpr int kActMax = kDensityActMax;  // chunks per workgroup the activity table holds (beyond: every chunk counts as active)
  uint8_t* act = reinterpret_cast<uint8_t*>(w0s + H * INP + 2 * H);  // [kActMax], behind the staged weights
  const bool use_act = ray_mask != nullptr && per_wg <= kActMax;
  if (use_act) {
    for (int64_t k = threadIdx.x; k < per_wg; k += kMlpBlock) act[k] = 0;
    __syncthreads();
    bool any = false;
    if (c_lo < c_hi) {
      const int64_t p_first = c_lo * kMlpBlock, p_last = (c_hi * kMlpBlock < M ? c_hi * kMlpBlock : M) - 1;
      const int64_t r_first = p_first / spr, r_last = p_last / spr;
      for (int64_t r = r_first + threadIdx.x; r <= r_last; r += kMlpBlock) {
        if (ray_mask[r] != 0) {
          any = true;
          const int64_t q0 = r * spr > p_first ? r * spr : p_first, q1 = (r + 1) * spr - 1 < p_last ? (r + 1) * spr - 1 : p_last;
          for (int64_t c = q0 / kMlpBlock; c <= q1 / kMlpBlock; ++c) act[c - c_lo] = 1;  // (same value from every writer)
        }
      }
    }
    if (!__syncthreads_or(any)) {
      if (partials != nullptr) {
        float* row = partials + (size_t)blockIdx.x * density_partial_stride(IN, H);
        for (int e = threadIdx.x; e < H * IN + 2 * H + 1; e += kMlpBlock) row[e] = 0.0f;
      }
      return;  // (without a partial buffer the sums go out as atomics: nothing to add)
    }
  }
  for (int e = threadIdx.x; e < H * INP; e += kMlpBlock) {
    const int j = e / INP, k = e - j * INP;
    w0s[e] = k < IN ? mlp.W0[j * IN + k] : 0.0f;
  }
  bool w_finite = true;  // a non-finite weight turns a zero gradient into NaN: such a network never skips a chunk
  for (int e = threadIdx.x; e < H * INP; e += kMlpBlock) w_finite = w_finite && fabsf(w0s[e]) <= 3.4028234663852886e38f;
  for (int e = threadIdx.x; e < H; e += kMlpBlock) {
    b0s[e] = mlp.b0[e];
    w1s[e] = mlp.W1[e];
    w_finite = w_finite && fabsf(b0s[e]) <= 3.4028234663852886e38f && fabsf(w1s[e]) <= 3.4028234663852886e38f;
  }
  const bool may_skip = __syncthreads_and(w_finite) != 0;

  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NT = H / 16;  // row tiles of dW0
  static_assert(H % 16 == 0 && IN <= 16, "dW0 tiling");
  v4f accM[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) accM[n] = v4f{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, jj = lane & 15, gg = lane >> 4;
  float accV = 0.0f;  // thread t < H: db0[t]; H <= t < 2H: dW1[t-H]; t == 2H: db1

  // The chunk loop alternates a per-point phase (global loads) and a weight-gradient phase (LDS only): the next chunk's
  // inputs are fetched during the latter, otherwise every chunk starts with an exposed HBM round trip.
  float x_next[IN];
  float gd_next = 0.0f, sel_next = 1.0f, pre_next = 0.0f;
  // Gated call with a per-ray mask (nsamd_weights_bwd_gate): a chunk none of whose rays carries gradient is not even
  // loaded — its density gradients are exact zeros, and the table scatter, which takes the same mask, never reads its
  // `denc`. The predicate depends on the chunk only: uniform over the workgroup.
  bool act_next = true;
  auto chunk_active = [&](int64_t c) {
    if (ray_mask == nullptr) return true;
    if (use_act) return act[c - c_lo] != 0;
    const int64_t p0 = c * kMlpBlock, p1 = (p0 + kMlpBlock < M ? p0 + kMlpBlock : M) - 1;
    bool any = false;
    for (int64_t r = p0 / spr; r <= p1 / spr; ++r) any = any || ray_mask[r] != 0;
    return any;
  };
  auto fetch = [&](int64_t c) {
    const int64_t p = c * kMlpBlock + threadIdx.x;
    act_next = c < c_hi && chunk_active(c);
    const bool live = act_next && p < M;
#pragma unroll
    for (int k = 0; k < IN; ++k) x_next[k] = live ? enc[(int64_t)k * M + p] : 0.0f;
    gd_next = live ? ddensity[p] : 0.0f;
    sel_next = (live && selector) ? selector[p] : 1.0f;
    pre_next = live ? pre[p] : 0.0f;
  };
  fetch(c_lo);
  for (int64_t c = c_lo; c < c_hi; ++c) {
    if (!act_next) {  // no ray of this chunk carries gradient (workgroup-uniform): nothing to add, nothing to write
      fetch(c + 1);
      continue;
    }
    const int64_t p = c * kMlpBlock + threadIdx.x;
    const bool live = p < M;
    float x[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) x[k] = x_next[k];
    // d density / d pre = avg * sel * exp(clamp(pre, -15, 15))            (activations.py:39-42)
    const float g_pre =
        live ? gd_next * sel_next * mlp.average_init_density * expf(fminf(fmaxf(pre_next, -15.0f), 15.0f)) : 0.0f;
    // A chunk whose 256 points all have a zero upstream gradient (finite weights and features) adds exact zeros to
    // every sum: only its zero feature gradients are written (the table scatter reads them), nothing is recomputed.
    // `g_pre != 0` is true for NaN: a non-finite gradient is never skipped. (The interlevel loss reaches 1-2 % of the
    // second proposal level's samples, profiles/r02_study_proposal_sparsity.txt.)
    {
      bool carries = g_pre != 0.0f || !may_skip;
#pragma unroll
      for (int k = 0; k < IN; ++k) carries = carries || !(fabsf(x[k]) <= 3.4028234663852886e38f);
      if (!__syncthreads_or(carries)) {
        if (live) {
#pragma unroll
          for (int k = 0; k < IN; ++k) denc[(int64_t)k * M + p] = 0.0f;
        }
        fetch(c + 1);
        continue;
      }
    }
    float dx[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) dx[k] = 0.0f;
#pragma unroll 4
    for (int j = 0; j < H; ++j) {
      float wj[INP];
#pragma unroll
      for (int k4 = 0; k4 < INP; k4 += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(w0s + j * INP + k4);
        wj[k4] = w4.x; wj[k4 + 1] = w4.y; wj[k4 + 2] = w4.z; wj[k4 + 3] = w4.w;
      }
      float a = b0s[j];
#pragma unroll
      for (int k = 0; k < IN; ++k) a = fmaf(wj[k], x[k], a);
      const float gh = (a > 0.0f) ? g_pre * w1s[j] : 0.0f;
      gh_T[j * LD + threadIdx.x] = gh;
      hg_T[j * LD + threadIdx.x] = fmaxf(a, 0.0f) * g_pre;
#pragma unroll
      for (int k = 0; k < IN; ++k) dx[k] = fmaf(gh, wj[k], dx[k]);
    }
#pragma unroll
    for (int k = 0; k < IN; ++k) {
      x_T[k * LD + threadIdx.x] = x[k];
      if (live) denc[(int64_t)k * M + p] = dx[k];
    }
    gp[threadIdx.x] = g_pre;
    __syncthreads();
    fetch(c + 1);
    // weight-gradient partial sums over the 256 points of this chunk: dW0[i][j] += sum_p gh[i][p] x[j][p]
    // (A lane (i = lane & 15, k = lane >> 4) = gh_T[i][p], B lane (j, k) = x_T[j][p], 4 points per MFMA)
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      const int pt = 64 * wv + 4 * q + gg;
      const float b = jj < IN ? x_T[jj * LD + pt] : 0.0f;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        accM[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(gh_T[(16 * n + jj) * LD + pt], b, accM[n], 0, 0, 0);
    }
    if (threadIdx.x < 2 * H + 1) {
      const float* a = (threadIdx.x < H)       ? gh_T + threadIdx.x * LD
                       : (threadIdx.x < 2 * H) ? hg_T + (threadIdx.x - H) * LD
                                               : gp;
      float s = 0.0f;
#pragma unroll 8
      for (int q = 0; q < kMlpBlock; ++q) s += a[q];
      accV += s;
    }
    __syncthreads();
  }
  // accM lane (j, g) reg r = dW0[16n + 4g + r][j] of this wave's points: add the 4 waves up in LDS (free by now)
  float* red = lds;  // [H][16]
  for (int e = threadIdx.x; e < H * 16; e += kMlpBlock) red[e] = 0.0f;
  __syncthreads();
  for (int turn = 0; turn < kMlpBlock / 64; ++turn) {
    if (wv == turn) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(16 * n + 4 * gg + r) * 16 + jj] += accM[n][r];  // distinct address per lane
    }
    __syncthreads();
  }
  if (partials != nullptr) {
    // one row of partial sums per workgroup, [dW0 (H x IN) | db0 (H) | dW1 (H) | db1]: density_dw_reduce_kernel adds the
    // rows up in a fixed order, so the weight gradients are bit-reproducible (float atomics are not). (Folding that sum
    // into this launch — last workgroup to arrive — was measured: 58 -> 104 us, one 256-thread workgroup has 1/16 of the
    // follow-up launch's loads in flight; profiles/r03_negative_results.txt.)
    float* row = partials + (size_t)blockIdx.x * density_partial_stride(IN, H);
    for (int e = threadIdx.x; e < H * IN; e += kMlpBlock) row[e] = red[(e / IN) * 16 + (e % IN)];
    if (threadIdx.x < 2 * H + 1) row[H * IN + threadIdx.x] = accV;
    return;
  }
  for (int e = threadIdx.x; e < H * IN; e += kMlpBlock) unsafeAtomicAdd(dW0 + e, red[(e / IN) * 16 + (e % IN)]);
  if (threadIdx.x < H) unsafeAtomicAdd(db0 + threadIdx.x, accV);
  else if (threadIdx.x < 2 * H) unsafeAtomicAdd(dW1 + (threadIdx.x - H), accV);
  else if (threadIdx.x == 2 * H) unsafeAtomicAdd(db1, accV);
}

template <int IN, int H>
__global__ __launch_bounds__(kMlpBlock) void density_mlp_bwd_kernel(
    const float* __restrict__ enc, const float* __restrict__ selector, const float* __restrict__ pre,
    const float* __restrict__ ddensity, int64_t M, nsamd_density_mlp mlp, float* __restrict__ denc,
    float* __restrict__ dW0, float* __restrict__ db0, float* __restrict__ dW1, float* __restrict__ db1,
    float* __restrict__ partials, const uint32_t* __restrict__ gate, const uint8_t* __restrict__ ray_mask, int spr) {
  density_mlp_bwd_body<IN, H>(enc, selector, pre, ddensity, M, mlp, denc, dW0, db0, dW1, db1, partials, gate, ray_mask, spr,
                              (int)gridDim.x);
}

// what one call of the kernel takes (proposal_chain.h: DensityBwdCall, plus the launch's choices)
struct DensityBwdArgs {
  const float* enc;
  const float* selector;
  const float* pre;
  const float* ddensity;
  int64_t M;
  nsamd_density_mlp mlp;
  float* denc;
  float* dW0;
  float* db0;
  float* dW1;
  float* db1;
  float* partials;
  const uint32_t* gate;
  const uint8_t* ray_mask;
  int spr;
  int nblocks;
};

// two independent calls (the two proposal levels of an update iteration) in one launch: blockIdx.y selects the call
template <int IN, int H>
__global__ __launch_bounds__(kMlpBlock) void density_mlp_bwd_pair_kernel(DensityBwdArgs a, DensityBwdArgs b) {
  if (blockIdx.y == 0) {
    if ((int)blockIdx.x >= a.nblocks) return;
    density_mlp_bwd_body<IN, H>(a.enc, a.selector, a.pre, a.ddensity, a.M, a.mlp, a.denc, a.dW0, a.db0, a.dW1, a.db1,
                                a.partials, a.gate, a.ray_mask, a.spr, a.nblocks);
  } else {
    if ((int)blockIdx.x >= b.nblocks) return;
    density_mlp_bwd_body<IN, H>(b.enc, b.selector, b.pre, b.ddensity, b.M, b.mlp, b.denc, b.dW0, b.db0, b.dW1, b.db1,
                                b.partials, b.gate, b.ray_mask, b.spr, b.nblocks);
  }
}

// grads += sum over the workgroups' partial rows, in a fixed order: 64 elements x 16 row-groups per workgroup, every
// thread has 16 loads in flight (the partials are a pure latency problem), the 16 group sums meet in LDS.
constexpr int kDwGroups = 16;
__device__ __forceinline__ void density_dw_reduce_body(const float* __restrict__ partials, int rows, int stride, int n_w0,
                                                       int hidden, float* __restrict__ dW0, float* __restrict__ db0,
                                                       float* __restrict__ dW1, float* __restrict__ db1,
                                                       const uint32_t* __restrict__ gate) {
  if (gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  __shared__ float part[kDwGroups][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int total = n_w0 + 2 * hidden + 1;
  float s = 0.f;
  if (e < total) {
    for (int b0i = grp; b0i < rows; b0i += kDwGroups * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        // (unconditional loads at a clamped row, dropped below: predicated, each load is waited for before the next is issued)
        const int b = b0i + u * kDwGroups;
        v[u] = partials[(size_t)(b < rows ? b : rows - 1) * stride + e];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (b0i + u * kDwGroups < rows) ? v[u] : 0.0f;
    }
  }
  part[grp][el] = s;
  __syncthreads();
  if (grp == 0 && e < total) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < kDwGroups; ++g2) t += part[g2][el];
    float* dst = e < n_w0 ? dW0 + e : (e < n_w0 + hidden ? db0 + (e - n_w0) : (e < n_w0 + 2 * hidden ? dW1 + (e - n_w0 - hidden) : db1));
    *dst += t;
  }
}

__global__ __launch_bounds__(64 * kDwGroups) void density_dw_reduce_kernel(const float* __restrict__ partials, int rows,
                                                                            int stride, int n_w0, int hidden,
                                                                            float* __restrict__ dW0,
                                                                            float* __restrict__ db0,
                                                                            float* __restrict__ dW1,
                                                                            float* __restrict__ db1,
                                                                            const uint32_t* __restrict__ gate) {
  density_dw_reduce_body(partials, rows, stride, n_w0, hidden, dW0, db0, dW1, db1, gate);
}

__global__ __launch_bounds__(64 * kDwGroups) void density_dw_reduce_pair_kernel(DensityBwdArgs a, DensityBwdArgs b, int stride,
                                                                                 int n_w0, int hidden) {
  if (blockIdx.y == 0) density_dw_reduce_body(a.partials, a.nblocks, stride, n_w0, hidden, a.dW0, a.db0, a.dW1, a.db1, a.gate);
  else density_dw_reduce_body(b.partials, b.nblocks, stride, n_w0, hidden, b.dW0, b.db0, b.dW1, b.db1, b.gate);
}

template <int IN, int H>
static int launch_fwd(const float* enc, const float* selector, int64_t M, nsamd_density_mlp mlp, float* density,
                      float* pre, hipStream_t stream) {
  const unsigned blocks = (unsigned)min((int64_t)8192, (M + kMlpBlock - 1) / kMlpBlock);
  density_mlp_fwd_kernel<IN, H><<<blocks, kMlpBlock, 0, stream>>>(enc, selector, M, mlp, density, pre);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

template <int IN, int H>
static int launch_bwd(const float* enc, const float* selector, const float* pre, const float* ddensity, int64_t M,
                      nsamd_density_mlp mlp, float* denc, float* dW0, float* db0, float* dW1, float* db1,
                      float* workspace, int64_t workspace_floats, const uint32_t* gate, const uint8_t* ray_mask, int spr,
                      hipStream_t stream) {
  const unsigned blocks = (unsigned)min((int64_t)kMaxBlocks, (M + kMlpBlock - 1) / kMlpBlock);
  const size_t lds = sizeof(float) * ((size_t)(2 * H + IN + 1) * (kMlpBlock + 1) + 8 + (size_t)H * (((IN + 3) & ~3) + 2)) +
                     kDensityActMax;
  if (lds > 64 * 1024) {  // per-device opt-in; cheap enough to repeat
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&density_mlp_bwd_kernel<IN, H>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return NSAMD_ERR_LAUNCH;
  }
  constexpr int stride = density_partial_stride(IN, H);
  float* partials = (workspace != nullptr && workspace_floats >= (int64_t)blocks * stride) ? workspace : nullptr;
  density_mlp_bwd_kernel<IN, H><<<blocks, kMlpBlock, lds, stream>>>(enc, selector, pre, ddensity, M, mlp, denc, dW0,
                                                                   db0, dW1, db1, partials, gate, ray_mask, spr);
  NSAMD_CHECK_LAUNCH();
  if (partials != nullptr) {
    const int total = H * IN + 2 * H + 1;
    density_dw_reduce_kernel<<<(total + 63) / 64, 64 * kDwGroups, 0, stream>>>(partials, (int)blocks, stride, H * IN, H,
                                                                             dW0, db0, dW1, db1, gate);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}

template <int IN, int H>
static int launch_bwd_pair(const DensityBwdCall& a, const DensityBwdCall& b, hipStream_t stream) {
  const DensityBwdCall* c[2] = {&a, &b};
  const size_t lds = sizeof(float) * ((size_t)(2 * H + IN + 1) * (kMlpBlock + 1) + 8 + (size_t)H * (((IN + 3) & ~3) + 2)) +
                     kDensityActMax;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&density_mlp_bwd_pair_kernel<IN, H>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return NSAMD_ERR_LAUNCH;
  }
  constexpr int stride = density_partial_stride(IN, H);
  DensityBwdArgs k[2];
  unsigned gx = 0;
  for (int i = 0; i < 2; ++i) {
    const unsigned blocks = (unsigned)min((int64_t)kMaxBlocks, (c[i]->M + kMlpBlock - 1) / kMlpBlock);
    // (the merged launch is for the training step's calls, which bring the scratch of the fixed-order reduce)
    if (c[i]->workspace == nullptr || c[i]->workspace_floats < (int64_t)blocks * stride) return NSAMD_ERR_UNSUPPORTED;
    k[i] = DensityBwdArgs{c[i]->enc, c[i]->selector, c[i]->pre, c[i]->ddensity, c[i]->M, c[i]->mlp, c[i]->denc, c[i]->dW0,
                          c[i]->db0, c[i]->dW1, c[i]->db1, c[i]->workspace, c[i]->gate, c[i]->ray_mask,
                          c[i]->ray_mask ? c[i]->spr : 1, (int)blocks};
    gx = blocks > gx ? blocks : gx;
  }
  if (a.workspace == b.workspace || a.dW0 == b.dW0) return NSAMD_ERR_UNSUPPORTED;  // one network for both levels: in turn
  density_mlp_bwd_pair_kernel<IN, H><<<dim3(gx, 2u), kMlpBlock, lds, stream>>>(k[0], k[1]);
  NSAMD_CHECK_LAUNCH();
  const int total = H * IN + 2 * H + 1;
  density_dw_reduce_pair_kernel<<<dim3((total + 63) / 64, 2u), 64 * kDwGroups, 0, stream>>>(k[0], k[1], stride, H * IN, H);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

int density_bwd_launch_pair(const DensityBwdCall& a, const DensityBwdCall& b, hipStream_t stream) {
  if (a.mlp.in_dim != b.mlp.in_dim || a.mlp.hidden != b.mlp.hidden || a.M <= 0 || b.M <= 0) return NSAMD_ERR_UNSUPPORTED;
  if (a.mlp.in_dim == 10 && a.mlp.hidden == 16) return launch_bwd_pair<10, 16>(a, b, stream);
  if (a.mlp.in_dim == 16 && a.mlp.hidden == 16) return launch_bwd_pair<16, 16>(a, b, stream);
  if (a.mlp.in_dim == 10 && a.mlp.hidden == 64) return launch_bwd_pair<10, 64>(a, b, stream);
  if (a.mlp.in_dim == 16 && a.mlp.hidden == 64) return launch_bwd_pair<16, 64>(a, b, stream);
  return NSAMD_ERR_UNSUPPORTED;
}

}  // namespace nsamd

using namespace nsamd;

#define NSAMD_DENSITY_DISPATCH(CALL)                                  \
  if (mlp.in_dim == 10 && mlp.hidden == 16) return CALL(10, 16);      \
  if (mlp.in_dim == 16 && mlp.hidden == 16) return CALL(16, 16);      \
  if (mlp.in_dim == 10 && mlp.hidden == 64) return CALL(10, 64);      \
  if (mlp.in_dim == 16 && mlp.hidden == 64) return CALL(16, 64);      \
  return NSAMD_ERR_UNSUPPORTED;

extern "C" int nsamd_density_mlp_fwd(const float* enc, const float* selector, int64_t M, nsamd_density_mlp mlp,
                                     float* density, float* pre, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(enc && density && mlp.W0 && mlp.b0 && mlp.W1 && mlp.b1);
#define CALL(IN, H) launch_fwd<IN, H>(enc, selector, M, mlp, density, pre, (hipStream_t)stream)
  NSAMD_DENSITY_DISPATCH(CALL)
#undef CALL
}

extern "C" int nsamd_density_field_fwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                                       nsamd_grid grid, nsamd_density_mlp mlp, float* enc, float* selector, float* density,
                                       float* pre, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(table && density && mlp.W0 && mlp.b0 && mlp.W1 && mlp.b1);
  NSAMD_REQUIRE(transform >= 0 && transform <= 2);
  if (pts.positions == nullptr) {
    NSAMD_REQUIRE(pts.origins && pts.directions && pts.t_bins && pts.samples_per_ray > 0 && M % pts.samples_per_ray == 0);
  }
  if (grid.log2_table_size < 1 || grid.log2_table_size > 28) return NSAMD_ERR_UNSUPPORTED;
  if (mlp.in_dim != 2 * grid.num_levels) return NSAMD_ERR_INVALID_ARG;
  const int64_t nb = (M + kMlpBlock - 1) / kMlpBlock;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  const float2* t2 = reinterpret_cast<const float2*>(table);
#define NSAMD_DENSITY_FWD(L_, H_)                                                                                       \
  density_field_fwd_kernel<L_, H_><<<(unsigned)nb, kMlpBlock, 0, (hipStream_t)stream>>>(pts, M, transform, aabb, t2, grid, mlp, \
                                                                                      enc, selector, density, pre)
  if (grid.num_levels == 5 && mlp.hidden == 16) {
    NSAMD_DENSITY_FWD(5, 16);
  } else if (grid.num_levels == 8 && mlp.hidden == 16) {
    NSAMD_DENSITY_FWD(8, 16);
  } else if (grid.num_levels == 5 && mlp.hidden == 64) {
    NSAMD_DENSITY_FWD(5, 64);
  } else if (grid.num_levels == 8 && mlp.hidden == 64) {
    NSAMD_DENSITY_FWD(8, 64);
  } else {
    return NSAMD_ERR_UNSUPPORTED;  // callers fall back to nsamd_hashgrid_encode_fwd + nsamd_density_mlp_fwd
  }
#undef NSAMD_DENSITY_FWD
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_density_mlp_bwd(const float* enc, const float* selector, const float* pre,
                                     const float* ddensity, int64_t M, nsamd_density_mlp mlp, float* denc,
                                     float* dW0, float* db0, float* dW1, float* db1, float* workspace,
                                     int64_t workspace_floats, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(enc && pre && ddensity && denc && dW0 && db0 && dW1 && db1 && mlp.W0 && mlp.b0 && mlp.W1 && mlp.b1);
#define CALL(IN, H)                                                                                          \
  launch_bwd<IN, H>(enc, selector, pre, ddensity, M, mlp, denc, dW0, db0, dW1, db1, workspace, workspace_floats, \
                    nullptr, nullptr, 1, (hipStream_t)stream)
  NSAMD_DENSITY_DISPATCH(CALL)
#undef CALL
}

extern "C" int nsamd_density_mlp_bwd_gated(const float* enc, const float* selector, const float* pre,
                                           const float* ddensity, int64_t M, nsamd_density_mlp mlp, float* denc,
                                           float* dW0, float* db0, float* dW1, float* db1, float* workspace,
                                           int64_t workspace_floats, const uint32_t* gate, const uint8_t* ray_mask,
                                           int32_t samples_per_ray, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && gate != nullptr);
  NSAMD_REQUIRE(ray_mask == nullptr || (samples_per_ray > 0 && M % samples_per_ray == 0));
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(enc && pre && ddensity && denc && dW0 && db0 && dW1 && db1 && mlp.W0 && mlp.b0 && mlp.W1 && mlp.b1);
#define CALL(IN, H)                                                                                          \
  launch_bwd<IN, H>(enc, selector, pre, ddensity, M, mlp, denc, dW0, db0, dW1, db1, workspace, workspace_floats, \
                    gate, ray_mask, ray_mask ? samples_per_ray : 1, (hipStream_t)stream)
  NSAMD_DENSITY_DISPATCH(CALL)
#undef CALL
}
