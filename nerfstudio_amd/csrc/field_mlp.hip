// nerfacto main-field MLPs on the gfx950 matrix cores, fp32 in / fp32 accumulate (v_mfma_f32_16x16x4_f32).
// Reference: NerfactoField.get_density + get_outputs, /root/reference/nerfstudio/fields/nerfacto_field.py:203-310
// (base MLP 32->64->16, trunc_exp density, SH16 | geo15 | appearance32 -> 64 -> 64 -> 3 sigmoid), torch path.
//
// fp32 because the contract is 1e-4 RGB L-inf against the fp32 torch field; gfx950 has no TF32, and the f32 MFMA
// is an exact k-ordered fmaf chain at the f32 vector rate — what it buys over VALU is operand bandwidth: 2048 MACs
// per instruction from two VGPRs, so the weights can stream from LDS at 16 B per lane per 4 MFMAs.
//
// Chain layout. One wavefront owns a tile of 16 sample points. A vector of F features of those points lives in
// registers as X[t][r] (t < F/16, r < 4): lane (j = lane & 15, g = lane >> 4) holds feature 16t + 4g + r of point
// j. With Y^T = W X^T, MFMA step (t, r) takes  A = W[16n + j][16t + 4g + r]  and  B = X[t][r]; the C/D fragment of
// output tile n is then neuron 16n + 4g + r' of point j — the SAME layout, so a layer's accumulators are the
// next layer's B operands with no shuffle or LDS round trip. The K-order permutation this implies is folded into
// the weight fragments when a workgroup stages them in LDS (once; workgroups are persistent over tiles):
//   Wf[n][t][lane][r] = W[16n + j][16t + 4g + r]   -> one conflict-free ds_read_b128 per lane feeds 4 MFMAs.
// That is the forward kernel (4 waves per workgroup, fragments staged once, workgroups persistent over tiles). The
// backward kernel (further down) recomputes the forward per tile, runs the data-gradient GEMMs in the same chain
// layout with W^T operands, and needs points on the k axis for the weight-gradient GEMMs: each wave transposes its
// (Dout, X) tiles through an LDS scratch (row stride 80 floats = 16 mod 32 banks: conflict-free b32 column reads).
// Its first version gave every wave all 48 dW accumulator tiles (424 registers, one wave per SIMD, MFMA busy 33 %);
// the current one shares dW out over 8 waves by output tile — see "backward: cooperative over the workgroup".
//
// Head input slots (internal K order of head layer 0): [0,16) SH, 16 = density pre-activation (zero weight),
// [17,32) geo features = base outputs 1..15 in place, [32,64) appearance embedding. Logical column = slot for
// slot < 16, slot - 1 for slot >= 17.
#include <stdlib.h>

#include "common.h"
#include "field_reduce.h"
#include "scatter.h"
#include "wave.h"

NSAMD_PROBE_DEFINE(field)

namespace nsamd {

typedef float v4f __attribute__((ext_vector_type(4)));


constexpr int kWaves = 4;
constexpr int kFieldThreads = 64 * kWaves;
constexpr int kScratchLd = 20;                      // floats per scratch row: 16 points + 4 (row stride = 4 mod 8 words:
                                                    // conflict-free b32 column stores and b128 row reads)
constexpr int kScratchTile = 64 * kScratchLd;       // one 64-feature x 16-point tile
__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Wf[n][t][lane][r] = Wint[16n + j][16t + 4g + r]. Staging is split into a load half and a store half so that a
// workgroup has the loads of ALL five layers in flight at once (48 per thread): as a load-wait-store loop the staging cost
// 23 k clocks of a wave's 113 k (probe_field_clocks, round 2).
template <int NT, int KT, int THREADS>
__device__ __forceinline__ void stage_frag_load(float* v, const float* __restrict__ W, int n_real, int k_real, bool head0,
                                                int app_dim) {
  static_assert(THREADS % 256 == 0 && (NT * KT * 256) % THREADS == 0, "whole tiles per pass");
  const int r = threadIdx.x & 3, lane = (threadIdx.x >> 2) & 63;
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < NT * KT * 256 / THREADS; ++i) {
    const int tile = (threadIdx.x >> 8) + (THREADS / 256) * i;
    const int t = tile % KT, n = tile / KT;
    const int row = 16 * n + j, slot = 16 * t + 4 * g + r;
    const int col = head0 ? head0_col(slot, app_dim) : slot;
    v[i] = (row < n_real && col >= 0 && col < k_real) ? W[row * k_real + col] : 0.0f;
  }
}

template <int NT, int KT, int THREADS>
__device__ __forceinline__ void stage_frag_store(float* dst, const float* v) {
#pragma unroll
  for (int i = 0; i < NT * KT * 256 / THREADS; ++i) dst[i * THREADS + threadIdx.x] = v[i];
}

template <int THREADS>
__device__ __forceinline__ void stage_bias(float* dst, const float* __restrict__ b, int n_real, int n_pad) {
  for (int e = threadIdx.x; e < n_pad; e += THREADS) dst[e] = (e < n_real) ? b[e] : 0.0f;
}

// out[n] (+)= sum over input tiles; frag = Wf-style block for this layer: [NT][KT][64][4]
template <int NT, int KT>
__device__ __forceinline__ void chain_gemm(const float* frag, const v4f* in, v4f* out, int lane) {
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    v4f a[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) a[n] = *reinterpret_cast<const v4f*>(frag + ((n * KT + t) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int n = 0; n < NT; ++n) out[n] = mfma16(a[n][r], in[t][r], out[n]);
    }
  }
}

// one input tile `t` of a [NT][KT] fragment block: out[n] += W[16n + j][16t ..] . in
template <int NT, int KT>
__device__ __forceinline__ void chain_gemm_tile(const float* frag, int t, const v4f& in, v4f* out, int lane) {
  v4f a[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) a[n] = *reinterpret_cast<const v4f*>(frag + ((n * KT + t) * 64 + lane) * 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int n = 0; n < NT; ++n) out[n] = mfma16(a[n][r], in[r], out[n]);
  }
}

template <int NT>
__device__ __forceinline__ void load_bias(const float* bias, v4f* out, int g) {
#pragma unroll
  for (int n = 0; n < NT; ++n) out[n] = *reinterpret_cast<const v4f*>(bias + 16 * n + 4 * g);
}

template <int NT>
__device__ __forceinline__ void relu_tiles(v4f* x) {
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[n][r] = fmaxf(x[n][r], 0.0f);
}

// ray (direction / camera row) of point p: 32-bit division whenever both fit (a 64-bit divide is ~100 instructions)
__device__ __forceinline__ int64_t ray_of(int64_t p, int64_t dir_group) {
  return (((uint64_t)p | (uint64_t)dir_group) >> 32) ? p / dir_group : (int64_t)((uint32_t)p / (uint32_t)dir_group);
}

struct TileInputs {
  int64_t p;       // clamped point index of this lane
  int64_t ray;     // p / dir_group
  bool live;       // point index < M
  float sel;       // selector
  int64_t cam;     // camera index (or 0)
};

// Forward of one 16-point tile; keeps every activation the backward needs.
struct FieldActs {
  v4f enc[2];
  v4f h1[4];      // relu
  v4f o16[1];     // base output (pre-activation)
  v4f hin[4];     // head input slots
  v4f ha[4];      // relu
  v4f hb[4];      // relu
  v4f rgbp[1];    // rgb pre-sigmoid (rows 0..2 of tile 0)
};

// head-input tile 0: the 4 SH components 4g + r of this lane's group, from the 16 of the view direction
// (base_field.py:136-142: SH of (dir + 1) / 2). Selected by exact 0/1 blending (x * 1 + 0 + 0 + 0 = x): written as
// selects, the compiler forms a dynamically indexed 16-float stack array — a scratch store + load per tile.
__device__ __forceinline__ v4f sh_quad(float dx, float dy, float dz, int g) {
  float sh[16];
  sh4_components((dx + 1.0f) / 2.0f, (dy + 1.0f) / 2.0f, (dz + 1.0f) / 2.0f, sh);
  const float m0 = g == 0 ? 1.0f : 0.0f, m1 = g == 1 ? 1.0f : 0.0f, m2 = g == 2 ? 1.0f : 0.0f, m3 = g == 3 ? 1.0f : 0.0f;
  v4f out;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r] = ((sh[r] * m0 + sh[4 + r] * m1) + sh[8 + r] * m2) + sh[12 + r] * m3;
  return out;
}

// encoded features: feature-major [32][M]; lane needs features 16t + 4g + r of its point
// Element (feature 16 t + 4 g + r, point p) of the feature-major [32, M] matrix = row base (16 t + r) M — uniform over the
// wave: a scalar address — plus ONE per-lane 32-bit offset 4 g M + p. Written as (16 t + 4 g + r) M + p the compiler hoists
// sixteen 64-bit per-lane products out of the tile loop: 32 loop-invariant VGPRs in kernels that run at the 256-register
// limit (and spill). The host bounds M (kMaxFieldPoints) so that the offset fits 32 bits.
constexpr int64_t kMaxFieldPoints = (int64_t)1 << 26;

__device__ __forceinline__ uint32_t enc_lane_offset(int64_t M, int64_t p, int lane) {
  return (uint32_t)(4 * (lane >> 4)) * (uint32_t)M + (uint32_t)p;
}

// (the forward kernels keep the plain form: at 112 VGPRs / 4 waves per SIMD they have room for the hoisted addresses, and
//  with the scalar bases their allocation came out at 128 registers + spills)
__device__ __forceinline__ void load_enc_tile_fwd(const float* __restrict__ enc, int64_t M, int64_t p, int lane, v4f* out) {
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[t][r] = enc[(int64_t)(16 * t + 4 * g + r) * M + p];
}

__device__ __forceinline__ void load_enc_tile(const float* __restrict__ enc, int64_t M, int64_t p, int lane, v4f* out) {
  const uint32_t off = enc_lane_offset(M, p, lane);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[t][r] = (enc + (int64_t)(16 * t + r) * M)[off];
}

// Ray terms (nsamd_field_mlp.ray_terms, include/nsamd.h): 48 of head layer 0's 64 input slots — the SH block and the appearance
// row — are the same for every sample of a ray, and a 16-point tile lies inside one ray, so their share of the layer's
// pre-activation (plus the bias) is a per-ray vector computed once by field_ray_terms_kernel. RAYC kernels start head layer
// 0's accumulators from it and run the per-point GEMM over the geo tile alone: K = 16 instead of 64 — 16 of the layer's 64
// MFMAs forward, 16 of 48 in the data gradient, and its weight gradient shrinks to the geo columns (the 48 per-ray columns
// follow from the per-tile sums of dL/d(pre-activation), field_reduce.h). A.enc must hold the tile's encoded features
// (load_enc_tile) on entry; RAYC: A.ha the ray's terms.
template <bool RAYC = false>
__device__ __forceinline__ void field_forward_tile(const float* wf, const float* bias,
                                                   const float* __restrict__ directions,
                                                   const float* __restrict__ app_table,
                                                   const float* __restrict__ app_const, int64_t dir_group, int64_t M,
                                                   int app_dim, const TileInputs& ti, int lane, FieldActs& A,
                                                   int probe_slot = 63, bool base_only = false) {
  const int g = lane >> 4;
  load_bias<4>(bias + kBiasBase0, A.h1, g);
  chain_gemm<4, 2>(wf + kOffBase0, A.enc, A.h1, lane);
  relu_tiles<4>(A.h1);
  load_bias<1>(bias + kBiasBase1, A.o16, g);
  chain_gemm<1, 4>(wf + kOffBase1, A.h1, A.o16, lane);
  PROBE_STAMP(kWaves, probe_slot);
  if (base_only) return;  // density only (Field.density_fn: the head's 8 320 of 11 392 MACs per point are not needed)

  if (RAYC) {
    chain_gemm_tile<4, 4>(wf + kOffHead0, 1, A.o16[0], A.ha, lane);
    relu_tiles<4>(A.ha);
    load_bias<4>(bias + kBiasHead1, A.hb, g);
    chain_gemm<4, 4>(wf + kOffHead1, A.ha, A.hb, lane);
    relu_tiles<4>(A.hb);
    load_bias<1>(bias + kBiasHead2, A.rgbp, g);
    chain_gemm<1, 4>(wf + kOffHead2, A.hb, A.rgbp, lane);
    return;
  }
  // head input: SH of (dir + 1) / 2  (base_field.py:136-142), geo in place, appearance embedding
  {
    const float* d = directions + 3 * ti.ray;
    A.hin[0] = sh_quad(d[0], d[1], d[2], g);
    if (app_dim > 0) {
      const float* src = (app_table != nullptr) ? app_table + ti.cam * 32 : app_const;
      A.hin[2] = *reinterpret_cast<const v4f*>(src + 4 * g);
      A.hin[3] = *reinterpret_cast<const v4f*>(src + 16 + 4 * g);
    } else {
      A.hin[2] = v4f{0.f, 0.f, 0.f, 0.f};
      A.hin[3] = v4f{0.f, 0.f, 0.f, 0.f};
    }
  }
  A.hin[1] = A.o16[0];
  load_bias<4>(bias + kBiasHead0, A.ha, g);
  chain_gemm<4, 4>(wf + kOffHead0, A.hin, A.ha, lane);
  relu_tiles<4>(A.ha);
  load_bias<4>(bias + kBiasHead1, A.hb, g);
  chain_gemm<4, 4>(wf + kOffHead1, A.ha, A.hb, lane);
  relu_tiles<4>(A.hb);
  load_bias<1>(bias + kBiasHead2, A.rgbp, g);
  chain_gemm<1, 4>(wf + kOffHead2, A.hb, A.rgbp, lane);
}

__device__ __forceinline__ TileInputs tile_inputs(int64_t tile, int lane, int64_t M, const float* selector,
                                                  const int64_t* cams, int64_t dir_group) {
  TileInputs ti;
  const int64_t p = tile * 16 + (lane & 15);
  ti.live = p < M;
  ti.p = ti.live ? p : M - 1;
  ti.sel = selector ? selector[ti.p] : 1.0f;
  ti.ray = ray_of(ti.p, dir_group);
  ti.cam = cams ? cams[ti.ray] : 0;
  return ti;
}

template <int THREADS>
__device__ __forceinline__ void stage_all_fwd(float* wf, float* bias, const nsamd_field_mlp& mlp, int app_dim) {
  constexpr int U = 256 * 4 / THREADS;  // loads per thread of a 4-tile layer
  float v[12 * U];
  stage_frag_load<4, 2, THREADS>(v, mlp.base_W0, 64, 32, false, 0);
  stage_frag_load<1, 4, THREADS>(v + 2 * U, mlp.base_W1, 16, 64, false, 0);
  stage_frag_load<4, 4, THREADS>(v + 3 * U, mlp.head_W0, 64, 31 + app_dim, true, app_dim);
  stage_frag_load<4, 4, THREADS>(v + 7 * U, mlp.head_W1, 64, 64, false, 0);
  stage_frag_load<1, 4, THREADS>(v + 11 * U, mlp.head_W2, 3, 64, false, 0);
  stage_frag_store<4, 2, THREADS>(wf + kOffBase0, v);
  stage_frag_store<1, 4, THREADS>(wf + kOffBase1, v + 2 * U);
  stage_frag_store<4, 4, THREADS>(wf + kOffHead0, v + 3 * U);
  stage_frag_store<4, 4, THREADS>(wf + kOffHead1, v + 7 * U);
  stage_frag_store<1, 4, THREADS>(wf + kOffHead2, v + 11 * U);
  stage_bias<THREADS>(bias + kBiasBase0, mlp.base_b0, 64, 64);
  stage_bias<THREADS>(bias + kBiasBase1, mlp.base_b1, 16, 16);
  stage_bias<THREADS>(bias + kBiasHead0, mlp.head_b0, 64, 64);
  stage_bias<THREADS>(bias + kBiasHead1, mlp.head_b1, 64, 64);
  stage_bias<THREADS>(bias + kBiasHead2, mlp.head_b2, 3, 16);
}

template <int WAVES, bool RAYC = false>
__global__ __launch_bounds__(64 * WAVES, WAVES == 16 ? 1 : 2) void field_mlp_fwd_kernel(
    const float* __restrict__ enc, const float* __restrict__ selector, const float* __restrict__ directions,
    const int64_t* __restrict__ cams, const float* __restrict__ app_const, int64_t dir_group, int64_t M,
    nsamd_field_mlp mlp, int app_dim, float* __restrict__ density, float* __restrict__ rgb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wf = lds;
  float* bias = lds + kFragTotal;
  PROBE_STAMP(WAVES, 0);
  stage_all_fwd<64 * WAVES>(wf, bias, mlp, app_dim);
  __syncthreads();
  PROBE_STAMP(WAVES, 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tiles = (M + 15) / 16;
  const float* app_table = cams ? mlp.appearance : nullptr;
  int probe_it = 0;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < tiles; tile += (int64_t)gridDim.x * WAVES) {
    // The fragments are loop-invariant LDS data: without this the compiler hoists all 192 VGPRs of them out of the
    // tile loop and the kernel drops to one wave per SIMD with nothing to hide the enc loads behind.
    asm volatile("" ::: "memory");
    const TileInputs ti = tile_inputs(tile, lane, M, selector, cams, dir_group);
    FieldActs A;
    load_enc_tile_fwd(enc, M, ti.p, lane, A.enc);
    if (RAYC && rgb != nullptr) {  // the ray's terms: in flight with the features, consumed two layers further down
      const float* c = mlp.ray_terms + ti.ray * 64 + 4 * (lane >> 4);
#pragma unroll
      for (int n = 0; n < 4; ++n) A.ha[n] = *reinterpret_cast<const v4f*>(c + 16 * n);
    }
    PROBE_STAMP(WAVES, 2 + 4 * probe_it);
    field_forward_tile<RAYC>(wf, bias, directions, app_table, app_const, dir_group, M, app_dim, ti, lane, A, 3 + 4 * probe_it,
                             rgb == nullptr);
    PROBE_STAMP(WAVES, 4 + 4 * probe_it);
    if (lane < 16 && ti.live) {  // g == 0 holds neurons 0..3 of tile 0
      density[ti.p] = mlp.average_init_density * expf(A.o16[0][0]) * ti.sel;
      if (rgb != nullptr) {
        float* o = rgb + 3 * ti.p;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = 1.0f / (1.0f + expf(-A.rgbp[0][c]));
      }
    }
    PROBE_STAMP(WAVES, 5 + 4 * probe_it);
    ++probe_it;
  }
  PROBE_STAMP(WAVES, 63);
}

// ray_terms [num_rays, 64] (+ ray_inputs [num_rays, 16 + 32]): head layer 0 on the 48 per-ray inputs, 16 RAYS per wavefront in
// the chain layout (a ray where the field kernels have a point): bias, then the SH tile, then the two appearance tiles.
__global__ __launch_bounds__(kFieldThreads) void field_ray_terms_kernel(
    const float* __restrict__ directions, const int64_t* __restrict__ cams, const float* __restrict__ app_const,
    int64_t num_rays, nsamd_field_mlp mlp, int app_dim, float* __restrict__ terms, float* __restrict__ inputs) {
  __shared__ __attribute__((aligned(16))) float wf[kFragHead0];
  __shared__ __attribute__((aligned(16))) float bias[64];
  {
    float v[16];
    stage_frag_load<4, 4, kFieldThreads>(v, mlp.head_W0, 64, 31 + app_dim, true, app_dim);
    stage_frag_store<4, 4, kFieldThreads>(wf, v);
    stage_bias<kFieldThreads>(bias, mlp.head_b0, 64, 64);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const float* app_table = cams ? mlp.appearance : nullptr;
  const int64_t tiles = (num_rays + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kWaves) {
    const int64_t ray = tile * 16 + j;
    const bool live = ray < num_rays;
    const int64_t rc = live ? ray : num_rays - 1;
    const float* d = directions + 3 * rc;
    v4f hin[4];
    hin[0] = sh_quad(d[0], d[1], d[2], g);
    if (app_dim > 0) {
      const float* src = (app_table != nullptr) ? app_table + cams[rc] * 32 : app_const;
      hin[2] = *reinterpret_cast<const v4f*>(src + 4 * g);
      hin[3] = *reinterpret_cast<const v4f*>(src + 16 + 4 * g);
    } else {
      hin[2] = v4f{0.f, 0.f, 0.f, 0.f};
      hin[3] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    v4f acc[4];
    load_bias<4>(bias, acc, g);
    chain_gemm_tile<4, 4>(wf, 0, hin[0], acc, lane);
    chain_gemm_tile<4, 4>(wf, 2, hin[2], acc, lane);
    chain_gemm_tile<4, 4>(wf, 3, hin[3], acc, lane);
    if (live) {
#pragma unroll
      for (int n = 0; n < 4; ++n) *reinterpret_cast<v4f*>(terms + ray * 64 + 16 * n + 4 * g) = acc[n];
      if (inputs != nullptr) {
        float* x = inputs + ray * (16 + app_dim);
        *reinterpret_cast<v4f*>(x + 4 * g) = hin[0];
        if (app_dim > 0) {
          *reinterpret_cast<v4f*>(x + 16 + 4 * g) = hin[2];
          *reinterpret_cast<v4f*>(x + 32 + 4 * g) = hin[3];
        }
      }
    }
  }
}

// ---- the bf16 matrix cores (v_mfma_f32_16x16x32_bf16: 8192 MACs in 4 passes, 16x the f32 MFMA's rate) ------------------------
// Used by the backward's weight-gradient GEMMs on two-piece operands (coop_dw_pk). Operand layout: lane (j, g) holds the 8
// contraction slots 8g .. 8g + 7 of row / column j, packed in pairs; result layout as the f32 MFMA's (probe_mfma_bf16_kernel,
// tests/test_gpu_kernels.py). (Round 2's forward on three-piece operands — fp32-accurate, 12 % faster, not bit-identical to the
// backward's recomputation — is csrc/experiments/rounds2to5_opt_in_variants.patch.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));  // 8 packed bf16

__device__ __forceinline__ v4f mfma_bf16(const u4& a, const u4& b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
  bf16x2 p;
  p[0] = (__bf16)a;
  p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, p);
}

// one bf16 MFMA with the assumed lane mapping (layout probe for the tests): A[16][32], B[32][16] row-major fp32 holding
// bf16-representable values
__global__ void probe_mfma_bf16_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  u4 a, b;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q] = pack_bf16(A[j * 32 + 8 * g + 2 * q], A[j * 32 + 8 * g + 2 * q + 1]);
    b[q] = pack_bf16(B[(8 * g + 2 * q) * 16 + j], B[(8 * g + 2 * q + 1) * 16 + j]);
  }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = mfma_bf16(a, b, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + j] = c[r];
}

// ---- backward -------------------------------------------------------------------------------------------------
// Attribution builds (scripts/build_variant.sh x "-DNSAMD_FIELD_BWD_SKIP_CONST=n", timing only, results are wrong): the bits of
// NSAMD_FIELD_BWD_SKIP as a compile-time constant in the PRODUCT kernel (the instrumented build's run-time switches cost
// registers and scalar spills of their own), plus 64 = no scratch stores (and splits), 128 = no forward GEMMs.
#ifndef NSAMD_FIELD_BWD_SKIP_CONST
#define NSAMD_FIELD_BWD_SKIP_CONST 0
#endif
constexpr int kSkipConst = NSAMD_FIELD_BWD_SKIP_CONST;

// store a chain-layout vector (T tiles of 16 features) as S[feature][point]: lane (j, g) register (t, r) is feature
// 16t + 4g + r of point j. Feature-major, so that a weight-gradient MFMA operand — 4 consecutive POINTS of one feature —
// is one ds_read_b128 (the point-major layout of round 1 cost one ds_read_b32 per MFMA operand, 2-3 LDS round trips per
// 1-2 MFMAs, r02b: dW 54 us against 31 us of MFMA time).
//
// Round 6: the scratch holds every value as TWO bf16 pieces in its dword — high half h = bf16(x) (RNE), low half
// m = bf16(x - h) (the residual is exact in fp32), x = h + m up to 2^-17 |x| — because the weight-gradient GEMMs run on the
// bf16 matrix cores (coop_dw). Same addresses, same one ds_write_b32 per value as the fp32 scratch; 3.5 vector
// instructions per value for the split (one v_cvt_pk_bf16_f32 forms the h of two values).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& o0, unsigned& o1) {
  const unsigned hh = pack_bf16(x0, x1);                              // low = h0, high = h1
  const float r0 = x0 - __uint_as_float(hh << 16);                    // exact
  const float r1 = x1 - __uint_as_float(hh & 0xffff0000u);            // exact
  o0 = pack_bf16(r0, x0);                                             // low = m0, high = h0 (the same RNE as above)
  o1 = pack_bf16(r1, x1);
}

// (head layer 0 keeps fp32 scratch rows and the f32 weight-gradient GEMM: 48 of its 64 input slots — SH of the view direction,
//  appearance row — are constant over a ray, so the 2^-18 representation error of a two-piece value is the SAME in every sample
//  of the ray and does not average out over the points: 2.1 - 2.4e-6 relative L2 on that layer against 2 - 5e-7 on the other
//  four, scripts/study_bf16_wgrad.py; the bench-size float64 test allows 2e-6.)
template <int T>
__device__ __forceinline__ void store_rows_f32(float* S, const v4f* x, int j, int g) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(16 * t + 4 * g + r) * kScratchLd + j] = x[t][r];
}

template <int T>
__device__ __forceinline__ void store_rows_pk(float* S, const v4f* x, int j, int g) {
  unsigned* U = reinterpret_cast<unsigned*>(S);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      unsigned o0, o1;
      split_pair(x[t][r], x[t][r + 1], o0, o1);
      U[(16 * t + 4 * g + r) * kScratchLd + j] = o0;
      U[(16 * t + 4 * g + r + 1) * kScratchLd + j] = o1;
    }
}

template <int N>
__device__ __forceinline__ void zero_tiles(v4f* x) {
#pragma unroll
  for (int n = 0; n < N; ++n) x[n] = v4f{0.f, 0.f, 0.f, 0.f};
}

template <int NT>
__device__ __forceinline__ void relu_mask(v4f* grad, const v4f* act) {
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) grad[n][r] = (act[n][r] > 0.0f) ? grad[n][r] : 0.0f;
}

// ---- backward: cooperative over the workgroup --------------------------------------------------------------------
// 8 waves per workgroup = two per SIMD, so a wave's LDS/MFMA latencies are covered by its SIMD mate. What makes that
// fit (256 registers per wave, 160 KiB LDS):
//  * ONE copy of the weights in LDS, row-major with a row stride = 4 mod 32 floats. The forward A operand
//    W[16n+j][16t+4g .. +3] is one conflict-free ds_read_b128; the transposed operand of the data-gradient GEMMs,
//    W[16t+4g+r][16m+j], is four ds_read_b32 of the same array (2-way bank conflict) instead of a second 48 KiB copy.
//  * The weight-gradient GEMMs are shared out over the waves BY OUTPUT TILE instead of by point tile: every wave
//    stores the transposed (Dout, X) tiles of its 16 points in its scratch, the workgroup synchronises, and wave w
//    accumulates its own 1-2 tiles of dW over all 128 points of the 8 scratch areas. 7 accumulator tiles per wave
//    instead of 48, no cross-wave reduction at the end, and every wave writes its tiles of the partial straight out.
constexpr int kRayS = 16 * kScratchLd, kRayX = kRayS + 64;  // RAYC: S_tile [64] and the ray's inputs [48] behind the geo tile's 16 rows
constexpr int kCoopWaves = 8;
constexpr int kCoopThreads = 64 * kCoopWaves;
constexpr int kLd64 = 68, kLd32 = 36;  // row strides (floats) of the K = 64 / K = 32 weight matrices in LDS
constexpr int kRowBase0 = 0, kRowBase1 = kRowBase0 + 64 * kLd32, kRowHead0 = kRowBase1 + 16 * kLd64,
              kRowHead1 = kRowHead0 + 64 * kLd64, kRowHead2 = kRowHead1 + 64 * kLd64,
              kRowTotal = kRowHead2 + 16 * kLd64;  // 13184 floats = 51.5 KiB

// W (n_real x k_real, row-major in global memory) -> LDS rows [n_pad][ld], internal slot order, zero padded. Load half
// and store half, as in the forward: all 24 loads of a thread are in flight before the first LDS store.
template <int N_PAD, int K_PAD>
__device__ __forceinline__ void stage_rows_load(float* v, const float* __restrict__ W, int n_real, int k_real, bool head0,
                                                int app_dim) {
#pragma unroll
  for (int i = 0; i < N_PAD * K_PAD / kCoopThreads; ++i) {
    const int e = threadIdx.x + i * kCoopThreads;
    const int row = e / K_PAD, slot = e % K_PAD;
    const int col = head0 ? head0_col(slot, app_dim) : slot;
    v[i] = (row < n_real && col >= 0 && col < k_real) ? W[row * k_real + col] : 0.0f;
  }
}

template <int N_PAD, int K_PAD, int LD>
__device__ __forceinline__ void stage_rows_store(float* dst, const float* v) {
#pragma unroll
  for (int i = 0; i < N_PAD * K_PAD / kCoopThreads; ++i) {
    const int e = threadIdx.x + i * kCoopThreads;
    dst[(e / K_PAD) * LD + e % K_PAD] = v[i];
  }
}

// out[n] += W[16n + j][:] . in   (forward direction; Wrows = LDS rows with stride LD)
template <int NT, int KT, int LD>
__device__ __forceinline__ void rows_gemm_fwd(const float* Wrows, const v4f* in, v4f* out, int j, int g) {
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (kSkipConst & 128) continue;
    v4f a[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) a[n] = *reinterpret_cast<const v4f*>(Wrows + (16 * n + j) * LD + 16 * t + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int n = 0; n < NT; ++n) out[n] = mfma16(a[n][r], in[t][r], out[n]);
    }
  }
}

// out[m] += W[:][16m + j]^T . in   (data-gradient direction: MT input-feature tiles, NT neuron tiles). The 4 x MT
// transposed operands of a neuron tile are fetched together (ds_read_b32 each: one copy of the weights serves both
// directions) and then feed 4 x MT MFMAs: one LDS wait per 16 MFMAs instead of one per 4.
template <int MT, int NT, int LD>
__device__ __forceinline__ void rows_gemm_bwd(const float* Wrows, const v4f* in, v4f* out, int j, int g) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float a[4][MT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) a[r][m] = Wrows[(16 * t + 4 * g + r) * LD + 16 * m + j];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) out[m] = mfma16(a[r][m], in[t][r], out[m]);
  }
}

struct NoBetween {
  template <int I>
  __device__ __forceinline__ void at() const {}
};

// `between.at<I>()` runs after GEMM I (0..4) of the chain: the producer mode of the backward slots its record stores there
template <class Between = NoBetween, bool RAYC = false>
__device__ __forceinline__ void coop_forward_tile(const float* W, const float* bias, const float (&dir)[3],
                                                  const float* __restrict__ app_table,
                                                  const float* __restrict__ app_const, int app_dim,
                                                  const TileInputs& ti, int lane, FieldActs& A, const v4f* app_pre = nullptr,
                                                  const Between& between = Between(), const v4f* ray_terms = nullptr) {
  const int j = lane & 15, g = lane >> 4;
  load_bias<4>(bias + kBiasBase0, A.h1, g);
  rows_gemm_fwd<4, 2, kLd32>(W + kRowBase0, A.enc, A.h1, j, g);
  between.template at<0>();
  relu_tiles<4>(A.h1);
  load_bias<1>(bias + kBiasBase1, A.o16, g);
  rows_gemm_fwd<1, 4, kLd64>(W + kRowBase1, A.h1, A.o16, j, g);
  between.template at<1>();
  if (RAYC) {  // head layer 0 from the ray's terms + the geo tile (see field_forward_tile)
#pragma unroll
    for (int n = 0; n < 4; ++n) A.ha[n] = ray_terms[n];
    rows_gemm_fwd<4, 1, kLd64>(W + kRowHead0 + 16, A.o16, A.ha, j, g);
  } else {
  A.hin[0] = sh_quad(dir[0], dir[1], dir[2], g);
  A.hin[1] = A.o16[0];
  if (app_pre != nullptr) {  // fetched by the caller ahead of other memory traffic
    A.hin[2] = app_pre[0];
    A.hin[3] = app_pre[1];
  } else if (app_dim > 0) {
    const float* src = (app_table != nullptr) ? app_table + ti.cam * 32 : app_const;
    A.hin[2] = *reinterpret_cast<const v4f*>(src + 4 * g);
    A.hin[3] = *reinterpret_cast<const v4f*>(src + 16 + 4 * g);
  } else {
    A.hin[2] = v4f{0.f, 0.f, 0.f, 0.f};
    A.hin[3] = v4f{0.f, 0.f, 0.f, 0.f};
  }
  load_bias<4>(bias + kBiasHead0, A.ha, g);
  rows_gemm_fwd<4, 4, kLd64>(W + kRowHead0, A.hin, A.ha, j, g);
  }
  between.template at<2>();
  relu_tiles<4>(A.ha);
  load_bias<4>(bias + kBiasHead1, A.hb, g);
  rows_gemm_fwd<4, 4, kLd64>(W + kRowHead1, A.ha, A.hb, j, g);
  between.template at<3>();
  relu_tiles<4>(A.hb);
  load_bias<1>(bias + kBiasHead2, A.rgbp, g);
  rows_gemm_fwd<1, 4, kLd64>(W + kRowHead2, A.hb, A.rgbp, j, g);
  between.template at<4>();
}

// dW tile (n, m) += sum over the points of scratch areas [first, first + count): Dout^T X on the bf16 matrix cores
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate). The contraction index of one instruction is the 16 points of an area x the two
// pieces of a value: lane (j, g) holds k-slots 8g .. 8g + 7 = (m, h) of points 4g .. 4g + 3 — one ds_read_b128 of the operand's
// scratch row, exactly the fp32 scratch's read. With A = (m, h) and B = (m', h') one instruction sums m m' + h h' over the 16
// points; with A's halves swapped (one v_alignbit per dword, shared by the wave's column tiles) the other sums h m' + m h':
// all four piece products, i.e. (h + m)(h' + m'), from 2 instructions of 16 clocks per (area, tile) where the f32 path
// issued 4 of 32. Against float64 the two-piece operands cost <= 5e-6 max|dW| (profiles/r02_study_bf16_wgrad.txt: the sum
// over 196 608 points averages the 2^-17 per-value error down); the per-point GEMMs (forward, data gradients) stay f32.
// Bias gradient: sum over points of h + m, one v_dot2c_f32_bf16 against (1, 1) per dword.
__device__ __forceinline__ float dot2_ones(unsigned a, float c) {
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(a), "v"(0x3f803f80u));
  return c;
}

// (fp32 scratch rows, v_mfma_f32_16x16x4_f32: MFMA step q of an area takes points 4g + q)
template <int TILES>
__device__ __forceinline__ void coop_dw_f32(v4f* acc, float* dbacc, bool want_db, const float* scratch, int first,
                                            int count, int n, int m0, int j, int g) {
#pragma unroll 2
  for (int area = first; area < first + count; ++area) {
    const float* Sd = scratch + area * 2 * kScratchTile;
    const float* Sx = Sd + kScratchTile;
    const v4f a = *reinterpret_cast<const v4f*>(Sd + (16 * n + j) * kScratchLd + 4 * g);
    v4f b[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) b[i] = *reinterpret_cast<const v4f*>(Sx + (16 * (m0 + i) + j) * kScratchLd + 4 * g);
    if (want_db) *dbacc += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TILES; ++i) acc[i] = mfma16(a[q], b[i][q], acc[i]);
  }
}

template <int TILES>
__device__ __forceinline__ void coop_dw_pk(v4f* acc, float* dbacc, bool want_db, const float* scratch, int first,
                                           int count, int n, int m0, int j, int g) {
#pragma unroll 2
  for (int area = first; area < first + count; ++area) {
    const float* Sd = scratch + area * 2 * kScratchTile;
    const float* Sx = Sd + kScratchTile;
    const u4 a = *reinterpret_cast<const u4*>(Sd + (16 * n + j) * kScratchLd + 4 * g);
    u4 b[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) b[i] = *reinterpret_cast<const u4*>(Sx + (16 * (m0 + i) + j) * kScratchLd + 4 * g);
    if (want_db) {
      float s = *dbacc;
#pragma unroll
      for (int q = 0; q < 4; ++q) s = dot2_ones(a[q], s);
      *dbacc = s;
    }
    u4 ar;
#pragma unroll
    for (int q = 0; q < 4; ++q) ar[q] = __builtin_amdgcn_alignbit(a[q], a[q], 16);
#pragma unroll
    for (int i = 0; i < TILES; ++i) acc[i] = mfma_bf16(a, b[i], acc[i]);
#pragma unroll
    for (int i = 0; i < TILES; ++i) acc[i] = mfma_bf16(ar, b[i], acc[i]);
  }
}

// Build-time choice of the weight-gradient arithmetic (same-box A/B, scripts/build_variant.sh): NSAMD_DW_BF16 = 0 every layer on
// the f32 matrix-core path (rounds 2 - 5), 1 (default) two-piece bf16 for every layer but head layer 0, 2 for all five.
#ifndef NSAMD_DW_BF16
#define NSAMD_DW_BF16 1
#endif
template <int T, bool HEAD0 = false>
__device__ __forceinline__ void store_rows(float* S, const v4f* x, int j, int g) {
  if (kSkipConst & 64) return;
  if (NSAMD_DW_BF16 == 0 || (NSAMD_DW_BF16 == 1 && HEAD0)) store_rows_f32<T>(S, x, j, g);
  else store_rows_pk<T>(S, x, j, g);
}

template <int TILES, bool HEAD0 = false>
__device__ __forceinline__ void coop_dw(v4f* acc, float* dbacc, bool want_db, const float* scratch, int first,
                                        int count, int n, int m0, int j, int g) {
  if (NSAMD_DW_BF16 == 0 || (NSAMD_DW_BF16 == 1 && HEAD0)) coop_dw_f32<TILES>(acc, dbacc, want_db, scratch, first, count, n, m0, j, g);
  else coop_dw_pk<TILES>(acc, dbacc, want_db, scratch, first, count, n, m0, j, g);
}

// acc lane (j, g) reg r of tile (n, m) = dW[16n + 4g + r][slot 16m + j]: to the workgroup's partial row (plain stores,
// this wave is the only writer) or, without a partial buffer, straight into the gradient with atomics
__device__ __forceinline__ void coop_emit(const v4f& acc, int n, int m, int k_pad, float* partial_layer,
                                          float* __restrict__ dst, int n_real, int k_real, bool head0, int app_dim,
                                          int j, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * n + 4 * g + r, slot = 16 * m + j;
    if (partial_layer != nullptr) {
      partial_layer[row * k_pad + slot] = acc[r];
    } else if (dst != nullptr) {
      const int col = head0 ? head0_col(slot, app_dim) : slot;
      if (row < n_real && col >= 0 && col < k_real) unsafeAtomicAdd(dst + row * k_real + col, acc[r]);
    }
  }
}

__device__ __forceinline__ void coop_emit_bias(float v, int n, int j, int g, float* partial_bias, float* __restrict__ dst,
                                               int n_real) {
  // lane (j, g) holds the partial of neuron 16n + j over the points = g mod 4: fold the four g first
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  if (g == 0) {
    if (partial_bias != nullptr) partial_bias[16 * n + j] = v;
    else if (dst != nullptr && 16 * n + j < n_real) unsafeAtomicAdd(dst + 16 * n + j, v);
  }
}


// ---- the scatter's pass 1 inside the backward ("producer" mode, nsamd_field_mlp_bwd_scatter) ---------------------------
// The data gradient of base layer 0 IS the encoded-feature gradient the table scatter routes: lane (j, g) holds, for point j,
// the two features of levels 8t + 2g and 8t + 2g + 1 — so instead of storing `denc` (25 MB), launching the route kernel and
// having it load the gradients again, recompute every cell and write the 201 MB of records in a memory-bound launch of its
// own (75-80 us), each lane derives the x-pair records of its four (point, level) pairs here and stores them straight into
// the tile queues, where the stores overlap the other waves' MFMA work. The workgroups are persistent, so a workgroup OWNS one
// static segment per (level, tile): slot = segment base + ds_add_rtn rank, the rank counters live in LDS for the whole
// launch (4 KiB), and no record needs a global atomic unless its segment is full (then: the tile's dynamic area, then the
// spill list — exactly the route kernel's fallbacks; scatter.hip's apply pass cannot tell the difference).
struct RouteArgs {
  nsamd_points P;
  int transform;
  nsamd_aabb box;
  nsamd_grid grid;
  ScatterGeom G;
  ScatterBufs buf;
};

constexpr int kProducerSegCap = 256;  // records per (tile, workgroup) static segment: scripts/study_fused_route_overflow.py
constexpr int kRouteLevels = 16;  // the main field's grid: 32 features = the K of base layer 0
constexpr int kRouteCnt = kRouteLevels * (1 << kProducerMaxLog2Bins);

// Everything the record emission reads, in LDS (filled once per workgroup from the kernel arguments): held as kernel
// arguments the ~60 scalars stay live across the whole tile loop and spill (98 SGPRs / 27 VGPRs spilled in the first build).
struct RouteLds {
  uint32_t cnt[kRouteCnt];          // [levels][bins] records of this workgroup per tile (rank counters)
  // per level, ONE 16-byte record: a lane reads the scale, the queue offset and the running maximum of its four levels
  // (8 t + 2 g + rr) through one address register and immediate offsets. As three arrays the compiler kept twelve
  // loop-invariant addresses per lane, spilled five of them, and every reload inside the tile loop was an
  // `s_waitcnt vmcnt(0)` — on this ISA stores count in vmcnt, so each one waited for the record stores in flight.
  struct Level {
    float scal;
    uint32_t loff;
    uint32_t lmax;  // max |gradient| bits
    uint32_t pad;
  } lv[kRouteLevels];
  nsamd_points P;
  nsamd_aabb box;
  int32_t transform, num_levels, log2_table_size, slice_log2, log2_bins;
  uint32_t seg_cap, static_end, level_cap, spill_cap, segs;
  uint4* queues;
  uint32_t* dyn_cursor;
  uint32_t* hdr;
  uint4* spill_rec;
  uint32_t* spill_tile;
  uint32_t* counts;
  float stash[kCoopWaves][11][64];  // per wave: a tile's 8 feature gradients + normalised position, until its records are out
};
constexpr int kRouteLdsWords = (sizeof(RouteLds) + 3) / 4;
static_assert(sizeof(float) * (kRowTotal + 256 + kCoopWaves * 2 * kScratchTile) + sizeof(RouteLds) <= 160 * 1024, "160 KiB of LDS per CU");

__device__ __forceinline__ void route_lds_init(RouteLds* L, const RouteArgs& R) {
  for (int e = threadIdx.x; e < kRouteCnt; e += kCoopThreads) L->cnt[e] = 0u;
  if (threadIdx.x < kRouteLevels) {
    L->lv[threadIdx.x].scal = R.grid.scalings[threadIdx.x];
    L->lv[threadIdx.x].loff = R.G.level_off[threadIdx.x];
    L->lv[threadIdx.x].lmax = 0u;
    L->lv[threadIdx.x].pad = 0u;
  }
  if (threadIdx.x == 0) {
    L->P = R.P;
    L->box = R.box;
    L->transform = R.transform;
    L->num_levels = R.grid.num_levels;
    L->log2_table_size = R.grid.log2_table_size;
    L->slice_log2 = R.G.slice_log2;
    L->log2_bins = R.G.log2_bins;
    L->seg_cap = R.G.seg_cap;
    L->static_end = R.G.segs * R.G.seg_cap;
    L->level_cap = R.G.level_cap[0];
    L->spill_cap = R.G.spill_cap;
    L->segs = R.G.segs;
    L->queues = R.buf.queues;
    L->dyn_cursor = R.buf.dyn_cursor;
    L->hdr = R.buf.hdr;
    L->spill_rec = R.buf.spill_rec;
    L->spill_tile = R.buf.spill_tile;
    L->counts = R.buf.counts;
  }
}

// cold path: a record that found no room in its segment nor in the tile's dynamic area (or a pair straddling two tiles) goes
// to the spill list — inline, one returning atomic per record: a CALL here would make every register that is live around
// the emission (the next tile's activations) a caller-saved spill. The list holds the worst case (write-only gradient).
__device__ __forceinline__ void route_spill(RouteLds* L, uint32_t tile, uint4 rec) {
  const uint32_t pos = atomicAdd(L->hdr + kHdrSpillCount, 1u);
  if (pos < L->spill_cap) {
    L->spill_rec[pos] = rec;
    L->spill_tile[pos] = tile;
  } else {
    atomicAdd(L->hdr + kHdrEvtLost, 1u);
  }
}


// Slow path of one record (pair q of a level): the pair straddles two tiles, or its static segment is full — the tile's
// dynamic area (one returning global atomic), then the spill list. Re-derives the record; reached by a handful of records
// per launch at most (scripts/study_fused_route_overflow.py), so all it must be is correct and out of the hot path's way.
__device__ __noinline__ void route_record_slow(RouteLds* L, float x, float y, float z, float g0, float g1, int level, int q,
                                               bool segment_full) {
  const uint32_t mask = (1u << L->log2_table_size) - 1u;
  const int sl = L->slice_log2, lb = L->log2_bins;
  const uint32_t local_mask = (1u << sl) - 1u;
  const Cell c = locate_cell(x, y, z, L->lv[level].scal);
  const PairHash h = pair_hash(c, q, mask);
  const uint32_t bin = h.ia >> sl, tile0 = (uint32_t)level << lb;
  const float bz = (q & 2) ? c.w[2] : 1.0f - c.w[2];
  const float by = (q & 1) ? c.w[1] : 1.0f - c.w[1];
  const float a0 = (g0 * bz) * by, a1 = (g1 * bz) * by;
  if ((h.ib >> sl) != bin) {
    const float omx = 1.0f - c.w[0];
    route_spill(L, tile0 + bin, make_uint4(__float_as_uint(a0 * omx), __float_as_uint(a1 * omx), 0u, h.ia & local_mask));
    route_spill(L, tile0 + (h.ib >> sl), make_uint4(__float_as_uint(a0 * c.w[0]), __float_as_uint(a1 * c.w[0]), 0u, h.ib & local_mask));
    return;
  }
  if (!segment_full) return;
  const uint4 rec = make_uint4(__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(c.w[0]),
                               (h.ia & local_mask) | ((h.ib & local_mask) << 14) | 0x80000000u);
  const uint32_t Q = L->level_cap, static_end = L->static_end;
  const uint32_t pos = atomicAdd(L->dyn_cursor + (tile0 + bin), 1u);
  if (pos < Q - static_end) rec_store(L->queues + (L->lv[level].loff + bin * Q + static_end + pos), rec);
  else route_spill(L, tile0 + bin, rec);
}

// The FOUR records of the lane's level K at once — the stash reads, the cell and the level's constants are shared, the four
// rank atomics are in flight together, the four stores leave back to back. (One record at a time was the schedule while the
// record stores were FLAT instructions — each held up the wave's next LDS wait for a full L2 round trip, so they had to be
// kept apart —; as global stores they are fire-and-forget, and what is left of a record is its chain of ~6 dependent LDS
// round trips, which four records now share.)
template <int K>
__device__ __forceinline__ void route_level(RouteLds* L, const float (*stash)[64], int lane, int probe_skip) {
  constexpr int t = K >> 1, rr = K & 1;
  const int g = lane >> 4;
  const int level = 8 * t + 2 * g + rr;
  const float g0 = stash[4 * t + 2 * rr][lane], g1 = stash[4 * t + 2 * rr + 1][lane];
  if (level >= L->num_levels || (g0 == 0.0f && g1 == 0.0f)) return;  // adding zero is a no-op (NaN != 0: kept); dead lanes hold zeros
  const float x = stash[8][lane], y = stash[9][lane], z = stash[10][lane];
  const Cell c = locate_cell(x, y, z, L->lv[level].scal);
  {  // integer compare of |bits|: a NaN or Inf wins and marks the level non-finite
    const uint32_t b0 = __float_as_uint(g0) & 0x7fffffffu, b1 = __float_as_uint(g1) & 0x7fffffffu;
    atomicMax(&L->lv[level].lmax, b0 > b1 ? b0 : b1);
  }
  const uint32_t mask = (1u << L->log2_table_size) - 1u;
  const int sl = L->slice_log2;
  const uint32_t local_mask = (1u << sl) - 1u;
  const uint32_t C = L->seg_cap;
  uint32_t* const cnt = L->cnt + ((uint32_t)level << L->log2_bins);
  PairHash h[4];
  uint32_t rank[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) h[q] = pair_hash(c, q, mask);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    rank[q] = (probe_skip & 16) ? (uint32_t)(threadIdx.x & 127) : atomicAdd(cnt + (h[q].ia >> sl), 1u);  // ds_add_rtn_u32
  uint4* const base = L->queues + (L->lv[level].loff + blockIdx.x * C);
  const uint32_t level_cap = L->level_cap;
  bool slow[4], full[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t bin = h[q].ia >> sl;
    const bool straddle = (h[q].ib >> sl) != bin;
    // autograd order ((g * wz) * wy) * wx; the x factor is applied by pass 2
    const float bz = (q & 2) ? c.w[2] : 1.0f - c.w[2];
    const float by = (q & 1) ? c.w[1] : 1.0f - c.w[1];
    // (a straddling pair has taken a rank like any other: its slot gets a record that adds zero, the two halves leave
    //  through the slow path)
    const uint4 rec = make_uint4(straddle ? 0u : __float_as_uint((g0 * bz) * by), straddle ? 0u : __float_as_uint((g1 * bz) * by),
                                 __float_as_uint(c.w[0]),
                                 (h[q].ia & local_mask) | ((straddle ? h[q].ia : h[q].ib) & local_mask) << 14 | 0x80000000u);
    full[q] = rank[q] >= C;
    slow[q] = straddle || full[q];
    // (record indices fit 32 bits: the plan checks queue_records < 2^31)
    if (!full[q] && !(probe_skip & 8)) rec_store(base + (bin * level_cap + rank[q]), rec);
    full[q] = full[q] && !straddle;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (slow[q]) route_record_slow(L, x, y, z, g0, g1, level, q, full[q]);
}

// every record of the stashed tile at once (after the last tile of a workgroup; emission of a tile normally rides on the next one)
__device__ __forceinline__ void route_flush(RouteLds* L, const float (*stash)[64], int lane, int probe_skip) {
  route_level<0>(L, stash, lane, probe_skip);
  route_level<1>(L, stash, lane, probe_skip);
  route_level<2>(L, stash, lane, probe_skip);
  route_level<3>(L, stash, lane, probe_skip);
}

// Raw position of this lane's point of a tile, in two halves: `route_issue_position` puts the loads out (one phase ahead of
// their use: behind the base-layer-1 barrier, consumed after base layer 0's data-gradient GEMM), `route_finish_position` forms
// the position. The loads are UNCONDITIONAL — the common layout (rays + bin edges, one ray per `dir_group` samples) reads its
// three arrays, any other layout reads a dummy address here and takes `load_position` in the second half: loads inside a
// branch would be waited for at the branch's end (see fetch_tile).
struct RawPosition {
  float o[3], d[3], t0, t1;
};

__device__ __forceinline__ bool route_ray_layout(const RouteLds* L, int64_t dir_group) {
  return L->P.positions == nullptr && (int64_t)L->P.samples_per_ray == dir_group;
}

__device__ __forceinline__ void route_issue_position(const RouteLds* L, int64_t tile, int64_t tiles, int64_t M,
                                                     int64_t dir_group, int lane, RawPosition& rp) {
  const int64_t pt = tile * 16 + (lane & 15);
  const bool rays = route_ray_layout(L, dir_group) && tile < tiles && pt < M;
  const int64_t p = rays ? pt : 0;
  const int64_t S = dir_group, ray = ray_of(p, S), s = p - ray * S;
  // (pointers that live in LDS: global_ptr, or these are flat loads — common.h)
  const NSAMD_GLOBAL_AS float* dummy = global_ptr(reinterpret_cast<const float*>(L->hdr));
  const NSAMD_GLOBAL_AS float* tb = rays ? global_ptr(L->P.t_bins) + ray * (S + 1) + s : dummy;
  const NSAMD_GLOBAL_AS float* o = rays ? global_ptr(L->P.origins) + 3 * ray : dummy;
  const NSAMD_GLOBAL_AS float* d = rays ? global_ptr(L->P.directions) + 3 * ray : dummy;
  rp.o[0] = o[0], rp.o[1] = o[1], rp.o[2] = o[2];
  rp.d[0] = d[0], rp.d[1] = d[1], rp.d[2] = d[2];
  rp.t0 = tb[0], rp.t1 = tb[1];
}

// false: no point
__device__ __forceinline__ bool route_finish_position(const RouteLds* L, const RawPosition& rp, int64_t tile, int64_t tiles,
                                                      int64_t M, int64_t dir_group, int lane, float& x, float& y, float& z) {
  const int64_t p = tile * 16 + (lane & 15);
  x = y = z = 0.0f;
  if (!(tile < tiles && p < M)) return false;
  if (!route_ray_layout(L, dir_group)) {
    load_position(L->P, p, x, y, z);
  } else {  // Frustums.get_positions (cameras/rays.py:50-59), the operations of common.h's load_position
    const float span = rp.t0 + rp.t1;
    x = rp.o[0] + rp.d[0] * span / 2.0f;
    y = rp.o[1] + rp.d[1] * span / 2.0f;
    z = rp.o[2] + rp.d[2] * span / 2.0f;
  }
  return true;
}

// Everything a tile's backward reads from global memory, fetched in one go (producer mode: right after the previous tile's
// base-layer-0 barrier and BEFORE that tile's records are stored — vector memory operations of a wave retire in order, so
// loads issued behind the 16 scattered record stores would wait for the whole burst to drain through the L2).
struct TileFetch {
  TileInputs ti;   // (sel, cam: RAW loads, see tile_fetched)
  v4f enc[2];
  float up[4];     // RAW drgb[0..2], ddensity of the lane's point
  float dir[3];
  v4f app[2];      // RAW appearance row slice
  v4f cterm[4];    // RAYC: the ray's terms (instead of dir / app / cam)
  float xin;       // RAYC: lane l < 16 + app_dim: input l of the ray (ray_inputs)
};

// Every load here is UNCONDITIONAL and nothing loaded is touched (no select, no copy) before `tile_fetched` runs at the top of
// the next iteration: a load inside a branch — even a uniform one on a null pointer — makes the compiler close the branch with
// `s_waitcnt vmcnt(0)`, which exposed the whole fetch (and, stores counting in vmcnt on this ISA, every record store in
// flight) right here instead of behind the base-layer-0 weight-gradient GEMM that follows. Absent inputs read a valid dummy
// address (`enc`) and are replaced by their constants in `tile_fetched`. The camera index is loaded first and waited for
// with the other loads in flight behind it (its appearance row is the one dependent load).
template <bool RAYC = false>
__device__ __forceinline__ void fetch_tile(TileFetch& f, int64_t tile, int64_t tiles, int lane, int64_t M,
                                           const float* __restrict__ enc, const float* __restrict__ selector,
                                           const float* __restrict__ directions, const int64_t* __restrict__ cams,
                                           const float* __restrict__ app_table, const float* __restrict__ app_const,
                                           int app_dim, int64_t dir_group, const float* __restrict__ ddensity,
                                           const float* __restrict__ drgb, const float* __restrict__ ray_terms = nullptr,
                                           const float* __restrict__ ray_inputs = nullptr) {
  const int g = lane >> 4;
  const int64_t t = tile < tiles ? tile : tiles - 1;
  const int64_t p = t * 16 + (lane & 15);
  f.ti.live = p < M && tile < tiles;  // (tile >= tiles: idle wave of the last round — computes, contributes zeros)
  f.ti.p = p < M ? p : M - 1;
  f.ti.ray = ray_of(f.ti.p, dir_group);
  if (RAYC) {
    f.ti.cam = 0;
  } else {
    const int64_t* cam_src = cams != nullptr ? cams + f.ti.ray : reinterpret_cast<const int64_t*>(enc);
    f.ti.cam = *cam_src;
  }
  f.ti.sel = (selector != nullptr ? selector : enc)[f.ti.p];
  load_enc_tile(enc, M, f.ti.p, lane, f.enc);
  // (single-dword loads kept apart by compiler barriers: merged into dwordx3 the triples land in register tuples that are
  //  re-shuffled with v_mov right behind the load — a wait for the whole fetch at the fetch site, the thing this avoids)
#define NSAMD_KEEP_APART() asm volatile("" ::: "memory")
  f.up[0] = drgb[3 * f.ti.p + 0];
  NSAMD_KEEP_APART();
  f.up[1] = drgb[3 * f.ti.p + 1];
  NSAMD_KEEP_APART();
  f.up[2] = drgb[3 * f.ti.p + 2];
  NSAMD_KEEP_APART();
  f.up[3] = ddensity[f.ti.p];
  if (RAYC) {
    const float* c = ray_terms + f.ti.ray * 64 + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) f.cterm[n] = *reinterpret_cast<const v4f*>(c + 16 * n);
    const int ncols = 16 + app_dim;
    f.xin = ray_inputs[f.ti.ray * ncols + (lane < ncols ? lane : 0)];
    return;
  }
  const float* d = directions + 3 * f.ti.ray;
  f.dir[0] = d[0];
  NSAMD_KEEP_APART();
  f.dir[1] = d[1];
  NSAMD_KEEP_APART();
  f.dir[2] = d[2];
  NSAMD_KEEP_APART();
  const float* src = app_table != nullptr ? app_table + (cams != nullptr ? f.ti.cam : 0) * 32
                                          : (app_const != nullptr ? app_const : enc);
  f.app[0] = *reinterpret_cast<const v4f*>(src + 4 * g);
  f.app[1] = *reinterpret_cast<const v4f*>(src + 16 + 4 * g);
#undef NSAMD_KEEP_APART
}

// the record stores that ride on a tile's forward recomputation: the previous tile's levels K = 0 (behind base layer 0's
// GEMM) and K = 1 (behind head layer 0's); K = 2, 3 leave between the backward phases
struct RouteBetween {
  RouteLds* L;
  const float (*stash)[64];
  int lane, probe_skip;
  bool on;
  template <int I>
  __device__ __forceinline__ void at() const {
    if (!on) return;
    if (I == 0) route_level<0>(L, stash, lane, probe_skip);
    if (I == 2) route_level<1>(L, stash, lane, probe_skip);
  }
};

// ROUTE: the kernel also emits the table scatter's pass-1 records. A tile's inputs are fetched one tile ahead.
// RAYC: head layer 0 from the ray terms (mlp.ray_terms; the host has checked that every tile lies inside one ray). The layer's
// per-point input is the geo tile alone, so its weight gradient's geo columns are formed like the other layers' (two-piece bf16:
// both operands vary from point to point). Its 48 per-ray columns and the appearance rows' gradient need only
// S_tile = the 64 sums of dL/d(pre-activation) over the tile's 16 points (DPP butterflies, fp32): with the 8 tiles of a
// workgroup iteration on the k axis, dW0[:, per-ray columns] += S^T X (X = the tiles' ray_inputs rows) is 24 f32 MFMAs per
// iteration for the whole workgroup where the per-point form took 384, and the tiles' appearance-gradient rows are
// W0[:, appearance]^T S (32 MFMAs) — written to `app_partials` exactly as the plain kernel writes its per-tile sums.
template <bool ROUTE, bool RAYC = false>
__global__ __launch_bounds__(kCoopThreads) void field_mlp_bwd_kernel(
    const float* __restrict__ enc, const float* __restrict__ selector, const float* __restrict__ directions,
    const int64_t* __restrict__ cams, const float* __restrict__ app_const, int64_t dir_group, int64_t M,
    nsamd_field_mlp mlp, int app_dim, const float* __restrict__ ddensity, const float* __restrict__ drgb,
    float* __restrict__ denc, nsamd_field_mlp_grads grads, float* __restrict__ partials,
    float* __restrict__ app_partials, int app_rows_per_point, int probe_skip_arg, RouteArgs R) {
  // probe_skip (NSAMD_FIELD_BWD_SKIP, timing experiments only — results are wrong when set): 1 = no weight-gradient
  // MFMAs, 2 = no workgroup barriers inside the tile loop, 4 = no data-gradient GEMMs. A run-time value only in the
  // instrumented build (`make probe`): the product kernel folds the switches away (a dozen scalar conditions and their
  // registers in a kernel whose scalar registers spill).
#ifdef NSAMD_PROBE_CLOCKS
  const int probe_skip = probe_skip_arg;
#else
  constexpr int probe_skip = kSkipConst;  // 0 in the product; -DNSAMD_FIELD_BWD_SKIP_CONST=n: attribution builds (wrong results)
  (void)probe_skip_arg;
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W = lds;                     // kRowTotal
  float* bias = lds + kRowTotal;      // 256
  float* scratch = bias + 256;        // kCoopWaves x 2 tiles
  RouteLds* RL = reinterpret_cast<RouteLds*>(scratch + kCoopWaves * 2 * kScratchTile);  // (ROUTE only: behind the scratch)
  if (ROUTE) route_lds_init(RL, R);
  PROBE_STAMP(kCoopWaves, 0);
  {
    float v[24];
    stage_rows_load<64, 32>(v, mlp.base_W0, 64, 32, false, 0);
    stage_rows_load<16, 64>(v + 4, mlp.base_W1, 16, 64, false, 0);
    stage_rows_load<64, 64>(v + 6, mlp.head_W0, 64, 31 + app_dim, true, app_dim);
    stage_rows_load<64, 64>(v + 14, mlp.head_W1, 64, 64, false, 0);
    stage_rows_load<16, 64>(v + 22, mlp.head_W2, 3, 64, false, 0);
    stage_rows_store<64, 32, kLd32>(W + kRowBase0, v);
    stage_rows_store<16, 64, kLd64>(W + kRowBase1, v + 4);
    stage_rows_store<64, 64, kLd64>(W + kRowHead0, v + 6);
    stage_rows_store<64, 64, kLd64>(W + kRowHead1, v + 14);
    stage_rows_store<16, 64, kLd64>(W + kRowHead2, v + 22);
  }
  for (int e = threadIdx.x; e < 256; e += kCoopThreads) {
    float v = 0.0f;
    if (e < kBiasBase1) v = mlp.base_b0[e];
    else if (e < kBiasHead0) v = mlp.base_b1[e - kBiasBase1];
    else if (e < kBiasHead1) v = mlp.head_b0[e - kBiasHead0];
    else if (e < kBiasHead2) v = mlp.head_b1[e - kBiasHead1];
    else if (e < kBiasHead2 + 3) v = mlp.head_b2[e - kBiasHead2];
    bias[e] = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  float* Sd = scratch + wave * 2 * kScratchTile;
  float* Sx = Sd + kScratchTile;
  const float* app_table = cams ? mlp.appearance : nullptr;
  // this wave's share of the weight gradients
  const int own_n = wave >> 1;           // 4 x 4 and 4 x 2 layers: row tile
  const int own_m2 = 2 * (wave & 1);     // 4 x 4 layers: column tiles own_m2, own_m2 + 1
  const int own_m1 = wave & 1;           // 4 x 2 layer: column tile
  const int own_q = wave & 3;            // 1 x 4 layers: column tile; the points are split in two halves
  const int own_half = wave >> 2;
  v4f dW_h1[2], dW_h0[2], dW_b0[1], dW_h2[1], dW_b1[1];
  v4f dW_r[2];  // RAYC: the per-ray columns of head layer 0: tile (row tile wave & 3, slot tile 0 | 2), waves 0..3 also (wave, slot tile 3)
  zero_tiles<2>(dW_r);
  float db_h1 = 0.f, db_h0 = 0.f, db_b0 = 0.f, db_h2 = 0.f, db_b1 = 0.f;
  zero_tiles<2>(dW_h1);
  zero_tiles<2>(dW_h0);
  zero_tiles<1>(dW_b0);
  zero_tiles<1>(dW_h2);
  zero_tiles<1>(dW_b1);
  const bool bias_owner44 = (wave & 1) == 0, bias_owner14 = own_q == 0;

  const int64_t tiles = (M + 15) / 16;
  const int64_t per_iter = (int64_t)gridDim.x * kCoopWaves;
  const int64_t iters = (tiles + per_iter - 1) / per_iter;
  PROBE_STAMP(kCoopWaves, 1);
  // the next tile's inputs are fetched behind the current tile's base-layer-0 barrier (NSAMD_NOROUTE_AHEAD=0 at build time:
  // only in the record-emitting variant — same-box A/B of the plain backward)
#ifndef NSAMD_NOROUTE_AHEAD
#define NSAMD_NOROUTE_AHEAD 1
#endif
  constexpr bool AHEAD = ROUTE || NSAMD_NOROUTE_AHEAD;
  TileFetch nxt;
  static_assert(AHEAD || !RAYC, "ray terms are fetched a tile ahead");
  if (AHEAD && iters > 0)
    fetch_tile<RAYC>(nxt, (int64_t)blockIdx.x * kCoopWaves + wave, tiles, lane, M, enc, selector, directions, cams, app_table,
                     app_const, app_dim, dir_group, ddensity, drgb, mlp.ray_terms, mlp.ray_inputs);
  for (int64_t it = 0; it < iters; ++it) {
    PROBE_STAMP(kCoopWaves, 2 + 10 * (int)it);
    const int64_t tile = (it * gridDim.x + blockIdx.x) * kCoopWaves + wave;
    TileInputs ti;
    FieldActs A;
    float up_rgb[3] = {0.f, 0.f, 0.f}, up_density = 0.f;
    float nxt_xin = 0.0f;
    if (AHEAD) {
      // (tile_fetched: what `fetch_tile` left raw gets its constants / masks here, where the values are first needed)
      ti = nxt.ti;
      if (selector == nullptr) ti.sel = 1.0f;
      if (cams == nullptr) ti.cam = 0;
      A.enc[0] = nxt.enc[0];
      A.enc[1] = nxt.enc[1];
      const bool mine = g == 0 && ti.live;
#pragma unroll
      for (int c = 0; c < 3; ++c) up_rgb[c] = mine ? nxt.up[c] : 0.0f;
      up_density = mine ? nxt.up[3] : 0.0f;
      const float dir[3] = {nxt.dir[0], nxt.dir[1], nxt.dir[2]};
      const v4f zero4 = v4f{0.f, 0.f, 0.f, 0.f};
      const v4f app[2] = {app_dim > 0 ? nxt.app[0] : zero4, app_dim > 0 ? nxt.app[1] : zero4};
      const v4f cterm[4] = {nxt.cterm[0], nxt.cterm[1], nxt.cterm[2], nxt.cterm[3]};
      nxt_xin = nxt.xin;
      if (ROUTE) {
        // records 0..6 of the PREVIOUS tile leave between this tile's forward GEMMs, the other nine between the phases below
        const RouteBetween rb{RL, RL->stash[wave], lane, probe_skip, it > 0 && !(probe_skip & 32)};
        coop_forward_tile<RouteBetween, RAYC>(W, bias, dir, app_table, app_const, app_dim, ti, lane, A, app, rb, cterm);
      } else {
        coop_forward_tile<NoBetween, RAYC>(W, bias, dir, app_table, app_const, app_dim, ti, lane, A, app, NoBetween{}, cterm);
      }
    } else {
      ti = tile_inputs(tile < tiles ? tile : tiles - 1, lane, M, selector, cams, dir_group);
      if (tile >= tiles) ti.live = false;  // idle wave of the last round: computes, contributes zeros
      load_enc_tile(enc, M, ti.p, lane, A.enc);
      // the upstream gradients of this tile (lanes g == 0 use them two and five phases further down): fetched with the
      // inputs, so their latency hides behind the forward instead of opening the head-2 and base-1 phases
      if (g == 0 && ti.live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) up_rgb[c] = drgb[3 * ti.p + c];
        up_density = ddensity[ti.p];
      }
      const float* d = directions + 3 * ti.ray;  // consumed two layers further down
      const float dir[3] = {d[0], d[1], d[2]};
      coop_forward_tile(W, bias, dir, app_table, app_const, app_dim, ti, lane, A);
    }
    PROBE_STAMP(kCoopWaves, 3 + 10 * (int)it);
    const bool emit = ROUTE && it > 0 && !(probe_skip & 32);  // the previous tile's records are still going out
#define NSAMD_ROUTE_LEVEL(K)                                                      \
  do {                                                                           \
    if (ROUTE) {                                                                 \
      if (emit) route_level<K>(RL, RL->stash[wave], lane, probe_skip);           \
    }                                                                            \
  } while (0)

    // ---- head layer 2 (64 -> 3, sigmoid) ----
    v4f g_rgbp[1];
    zero_tiles<1>(g_rgbp);
    if (g == 0 && ti.live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float sg = 1.0f / (1.0f + expf(-A.rgbp[0][c]));
        g_rgbp[0][c] = up_rgb[c] * (sg * (1.0f - sg));
      }
    }
    if (!(probe_skip & 2)) __syncthreads();  // the previous iteration's last weight-gradient reads of the scratch are done
    // Every layer: the wave stores its (Dout, X) tiles, runs its own data-gradient GEMM (registers + weights only) while
    // the stores drain, THEN meets the workgroup and takes its share of the weight gradient — so both barrier intervals
    // of a layer hold MFMA work (the store -> barrier interval used to hold none).
    store_rows<1>(Sd, g_rgbp, j, g);
    store_rows<4>(Sx, A.hb, j, g);
    v4f g_hb[4];
    zero_tiles<4>(g_hb);
    if (!(probe_skip & 4)) rows_gemm_bwd<4, 1, kLd64>(W + kRowHead2, g_rgbp, g_hb, j, g);
    relu_mask<4>(g_hb, A.hb);
    if (!(probe_skip & 2)) __syncthreads();
    if (!(probe_skip & 1)) coop_dw<1>(dW_h2, &db_h2, bias_owner14, scratch, 4 * own_half, 4, 0, own_q, j, g);
    if (!(probe_skip & 2)) __syncthreads();
    PROBE_STAMP(kCoopWaves, 4 + 10 * (int)it);

    // ---- head layer 1 (64 -> 64) ----
    store_rows<4>(Sd, g_hb, j, g);
    store_rows<4>(Sx, A.ha, j, g);
    v4f g_ha[4];
    zero_tiles<4>(g_ha);
    if (!(probe_skip & 4)) rows_gemm_bwd<4, 4, kLd64>(W + kRowHead1, g_hb, g_ha, j, g);
    relu_mask<4>(g_ha, A.ha);
    NSAMD_ROUTE_LEVEL(2);
    if (!(probe_skip & 2)) __syncthreads();
    if (!(probe_skip & 1)) coop_dw<2>(dW_h1, &db_h1, bias_owner44, scratch, 0, kCoopWaves, own_n, own_m2, j, g);
    if (!(probe_skip & 2)) __syncthreads();
    PROBE_STAMP(kCoopWaves, 5 + 10 * (int)it);

    // ---- head layer 0 (slots 64 -> 64) ----
    v4f g_hin[4];
    zero_tiles<4>(g_hin);
    if (RAYC) {
      // the geo tile is the only per-point input: X = base output tile (slot 16 = the density pre-activation: its column of the
      // partial row is dropped by the reduce), data gradient for that tile alone
      store_rows<4>(Sd, g_ha, j, g);
      store_rows<1>(Sx, A.o16, j, g);
      if (!(probe_skip & 4)) rows_gemm_bwd<1, 4, kLd64>(W + kRowHead0 + 16, g_ha, g_hin + 1, j, g);
      // S_tile: the tile's 64 sums of dL/d(pre-activation) over its 16 points (DPP butterfly, lane j == 0 of every row), and the
      // ray's inputs, into the free rows of the wave's X area (the geo tile takes 16 of its 64 rows)
      float* Ss = Sx + kRayS;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        v4f v = g_ha[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = row16_sum_lane0(v[r]);
        if (j == 0) *reinterpret_cast<v4f*>(Ss + 16 * n + 4 * g) = v;
      }
      if (lane < 48) Sx[kRayX + lane] = lane < 16 + app_dim ? nxt_xin : 0.0f;
      if (!(probe_skip & 2)) __syncthreads();
      if (!(probe_skip & 1)) {
        coop_dw<1>(dW_h0, &db_h0, true, scratch, 4 * own_half, 4, own_q, 0, j, g);
        // per-ray columns: k = the workgroup's 8 tiles (two instructions of 4), A = S (row 16 a + j), B = the tiles' inputs
        const int a = wave & 3, b0 = wave >> 2;  // b: 0 = SH, 1 / 2 = appearance 0..15 / 16..31
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float* area = scratch + (4 * h + g) * 2 * kScratchTile + kScratchTile;
          const float sa = area[kRayS + 16 * a + j];
          if (b0 == 0 || app_dim > 0) dW_r[0] = mfma16(sa, area[kRayX + 16 * b0 + j], dW_r[0]);
          if (wave < 4 && app_dim > 0) dW_r[1] = mfma16(sa, area[kRayX + 32 + j], dW_r[1]);
        }
        // the tiles' appearance-gradient rows: out[slot 16 w + j'][tile] = sum over n of W0[n][32 + slot] S_tile[n] (waves 0, 1)
        if (wave < 2 && app_partials != nullptr) {
          v4f o = {0.f, 0.f, 0.f, 0.f};
          const float* st = scratch + (j & 7) * 2 * kScratchTile + kScratchTile + kRayS;
#pragma unroll
          for (int kk = 0; kk < 16; ++kk)
            o = mfma16(W[kRowHead0 + (4 * kk + g) * kLd64 + 32 + 16 * wave + j], st[4 * kk + g], o);
          // lane (j, g): rows 4 g .. 4 g + 3 of slot tile `wave`, tile j of the workgroup's eight
          const int64_t tj = (it * gridDim.x + blockIdx.x) * kCoopWaves + j;
          if (j < 8 && tj < tiles)
            *reinterpret_cast<v4f*>(app_partials + ((uint32_t)tj * 32u + (uint32_t)(16 * wave + 4 * g))) = o;
        }
      }
      if (!(probe_skip & 2)) __syncthreads();
    } else {
    store_rows<4, true>(Sd, g_ha, j, g);
    store_rows<4, true>(Sx, A.hin, j, g);
    // input tile 0 is the SH block: it carries no gradient, so only columns 16..63 (tiles 1..3) are formed
    if (!(probe_skip & 4)) rows_gemm_bwd<3, 4, kLd64>(W + kRowHead0 + 16, g_ha, g_hin + 1, j, g);
    if (!(probe_skip & 2)) __syncthreads();
    if (!(probe_skip & 1)) coop_dw<2, true>(dW_h0, &db_h0, bias_owner44, scratch, 0, kCoopWaves, own_n, own_m2, j, g);
    if (!(probe_skip & 2)) __syncthreads();
    }
    PROBE_STAMP(kCoopWaves, 6 + 10 * (int)it);

    // appearance-embedding gradient (slots 32..63): rows of one camera are pre-reduced over the tile's points
    if (!RAYC && app_table != nullptr && grads.appearance != nullptr) {
      if (app_partials != nullptr && app_rows_per_point) {
        // per-sample cameras (packed instant-ngp samples, explicit positions: a 16-point tile spans several rays): every
        // point writes its own 32 gradients; the reduce launch adds the rows of each camera in point order — no atomics
        // (4.3 M float atomics per step on the instant-ngp workload: field_mlp_bwd 363 -> see profiles/r03_bench_ngp*)
        if (ti.live) {
#pragma unroll
          for (int t = 2; t < 4; ++t)
            *reinterpret_cast<v4f*>(app_partials + ti.p * 32 + 16 * (t - 2) + 4 * g) = g_hin[t];
        }
      } else if (app_partials != nullptr) {
        // every tile lies inside one ray (the host checked samples-per-ray % 16 == 0): its 32 sums go to a scratch row
        // and field_app_reduce_kernel adds the rows of each camera in a fixed order — bit-reproducible, no atomics
#pragma unroll
        for (int t = 2; t < 4; ++t) {
          v4f v = g_hin[t];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = row16_sum_lane0(v[r]);  // (DPP: the butterfly's sums for lane j == 0, wave.h)
          // (32-bit element offset from the kernel-argument base: as a hoisted 64-bit per-lane address this was a spilled
          //  register pair, reloaded here with `s_waitcnt vmcnt(0)` — a drain of every record store in flight per tile;
          //  tiles * 32 < 2^32 for any M whose 32 x M feature matrix exists)
          if (j == 0 && tile < tiles)
            *reinterpret_cast<v4f*>(app_partials + ((uint32_t)tile * 32u + (uint32_t)(16 * (t - 2) + 4 * g))) = v;
        }
      } else {
        const int cam32 = (int)ti.cam;
        const int cam0 = __shfl(cam32, 0);
        const bool uniform = __all(cam32 == cam0);
        if (uniform) {
#pragma unroll
          for (int t = 2; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = row16_sum_lane0(g_hin[t][r]);
              if (j == 0 && v != 0.0f) unsafeAtomicAdd(grads.appearance + (int64_t)cam0 * 32 + 16 * (t - 2) + 4 * g + r, v);
            }
        } else if (ti.live) {
#pragma unroll
          for (int t = 2; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              unsafeAtomicAdd(grads.appearance + ti.cam * 32 + 16 * (t - 2) + 4 * g + r, g_hin[t][r]);
        }
      }
    }

    // ---- base layer 1 (64 -> 16): slot 16 + n  <-  base output n; n = 0 is the density pre-activation ----
    v4f g_o16[1];
    g_o16[0] = g_hin[1];
    if (g == 0) {
      // d density / d pre = avg * sel * exp(clamp(pre,-15,15))   (activations.py:39-42); slot 16 has zero weight
      const float pre = A.o16[0][0];
      g_o16[0][0] = ti.live ? up_density * ti.sel * mlp.average_init_density *
                                  expf(fminf(fmaxf(pre, -15.0f), 15.0f))
                            : 0.0f;
    }
    store_rows<1>(Sd, g_o16, j, g);
    store_rows<4>(Sx, A.h1, j, g);
    v4f g_h1[4];
    zero_tiles<4>(g_h1);
    if (!(probe_skip & 4)) rows_gemm_bwd<4, 1, kLd64>(W + kRowBase1, g_o16, g_h1, j, g);
    relu_mask<4>(g_h1, A.h1);
    NSAMD_ROUTE_LEVEL(3);
    if (!(probe_skip & 2)) __syncthreads();
    if (!(probe_skip & 1)) coop_dw<1>(dW_b1, &db_b1, bias_owner14, scratch, 4 * own_half, 4, 0, own_q, j, g);
    if (!(probe_skip & 2)) __syncthreads();
    PROBE_STAMP(kCoopWaves, 7 + 10 * (int)it);

    // ---- base layer 0 (32 -> 64) ----
    RawPosition rawpos;
    if (ROUTE) route_issue_position(RL, tile, tiles, M, dir_group, lane, rawpos);
    store_rows<4>(Sd, g_h1, j, g);
    store_rows<2>(Sx, A.enc, j, g);
    v4f g_enc[2];
    zero_tiles<2>(g_enc);
    if (!(probe_skip & 4)) rows_gemm_bwd<2, 4, kLd32>(W + kRowBase0, g_h1, g_enc, j, g);
    if (ti.live && denc != nullptr) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) (denc + (int64_t)(16 * t + r) * M)[enc_lane_offset(M, ti.p, lane)] = g_enc[t][r];
    }
    // The scatter's pass-1 records (see RouteArgs / route_step): the last record of the PREVIOUS tile leaves, then this
    // tile's feature gradients and normalised positions take its place in the wave's stash; they go out one record at a time
    // during the next tile's iteration (after the last one: route_flush below).
    if (ROUTE) {  // (a wave reads only its own rows of the stash back: no barrier is involved)
      float px, py, pz;
      const bool plive = route_finish_position(RL, rawpos, tile, tiles, M, dir_group, lane, px, py, pz);
      (void)normalise_position(RL->transform, RL->box, px, py, pz);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) RL->stash[wave][4 * t + r][lane] = plive ? g_enc[t][r] : 0.0f;
      RL->stash[wave][8][lane] = px;
      RL->stash[wave][9][lane] = py;
      RL->stash[wave][10][lane] = pz;
    }
    if (!(probe_skip & 2)) __syncthreads();
    if (AHEAD && it + 1 < iters)  // the next tile's inputs (see TileFetch)
      fetch_tile<RAYC>(nxt, tile + per_iter, tiles, lane, M, enc, selector, directions, cams, app_table, app_const, app_dim,
                       dir_group, ddensity, drgb, mlp.ray_terms, mlp.ray_inputs);
    PROBE_STAMP(kCoopWaves, 9 + 10 * (int)it);
    if (!(probe_skip & 1)) coop_dw<1>(dW_b0, &db_b0, own_m1 == 0, scratch, 0, kCoopWaves, own_n, own_m1, j, g);
    // no barrier here: the next writer of the scratch is the next iteration's head layer 2, behind its own barrier
    PROBE_STAMP(kCoopWaves, 8 + 10 * (int)it);
#undef NSAMD_ROUTE_LEVEL
  }
  if (ROUTE && iters > 0 && !(probe_skip & 32)) route_flush(RL, RL->stash[wave], lane, probe_skip);  // the last tile's records
  PROBE_STAMP(kCoopWaves, 62);

  // ---- the two point-halves of the 1 x 4 layers meet in LDS (scratch is free now) ---------------------------------
  float* stash = scratch;  // [2 layers][4 tiles][64 lanes][4] + [2][4][64] bias partials
  __syncthreads();         // the last weight-gradient reads of the scratch are done (and every record has its rank)
  if (ROUTE) {  // this workgroup's segment counts and its share of the levels' gradient maxima
    const int B = 1 << RL->log2_bins;
    for (int e = threadIdx.x; e < RL->num_levels * B; e += kCoopThreads) {  // e = tile = (level << log2_bins) + bin
      const uint32_t n = RL->cnt[e];
      RL->counts[(size_t)e * RL->segs + blockIdx.x] = n < RL->seg_cap ? n : RL->seg_cap;
    }
    if (threadIdx.x < (unsigned)RL->num_levels && RL->lv[threadIdx.x].lmax != 0u)
      atomicMax(RL->hdr + threadIdx.x, RL->lv[threadIdx.x].lmax);
  }
  if (own_half == 1) {
    *reinterpret_cast<v4f*>(stash + ((0 * 4 + own_q) * 64 + lane) * 4) = dW_h2[0];
    *reinterpret_cast<v4f*>(stash + ((1 * 4 + own_q) * 64 + lane) * 4) = dW_b1[0];
    if (bias_owner14) {
      stash[2048 + lane] = db_h2;
      stash[2048 + 64 + lane] = db_b1;
    }
    if (RAYC) {  // head layer 0's geo columns: row tile own_q, the points split in two halves like the 1 x 4 layers
      *reinterpret_cast<v4f*>(stash + 2304 + (own_q * 64 + lane) * 4) = dW_h0[0];
      stash[2304 + 1024 + own_q * 64 + lane] = db_h0;
    }
  }
  __syncthreads();
  float* prow = partials != nullptr ? partials + (size_t)blockIdx.x * kPartialStride : nullptr;
  float* pbias = prow != nullptr ? prow + kFragTotal : nullptr;
  if (prow != nullptr) {  // padding of the row that no wave owns: the bias tail
    for (int e = kBiasTotal + threadIdx.x; e < 256; e += kCoopThreads) pbias[e] = 0.0f;
  }
  if (own_half == 0) {
    const v4f o_h2 = *reinterpret_cast<const v4f*>(stash + ((0 * 4 + own_q) * 64 + lane) * 4);
    const v4f o_b1 = *reinterpret_cast<const v4f*>(stash + ((1 * 4 + own_q) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { dW_h2[0][r] += o_h2[r]; dW_b1[0][r] += o_b1[r]; }
    coop_emit(dW_h2[0], 0, own_q, 64, prow ? prow + kOffHead2 : nullptr, grads.head_W2, 3, 64, false, 0, j, g);
    coop_emit(dW_b1[0], 0, own_q, 64, prow ? prow + kOffBase1 : nullptr, grads.base_W1, 16, 64, false, 0, j, g);
    if (bias_owner14) {
      coop_emit_bias(db_h2 + stash[2048 + lane], 0, j, g, pbias ? pbias + kBiasHead2 : nullptr, grads.head_b2, 3);
      coop_emit_bias(db_b1 + stash[2048 + 64 + lane], 0, j, g, pbias ? pbias + kBiasBase1 : nullptr, grads.base_b1, 16);
    }
    if (RAYC) {
      const v4f o_h0 = *reinterpret_cast<const v4f*>(stash + 2304 + (own_q * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) dW_h0[0][r] += o_h0[r];
      coop_emit(dW_h0[0], own_q, 1, 64, prow ? prow + kOffHead0 : nullptr, grads.head_W0, 64, 31 + app_dim, true, app_dim, j, g);
      coop_emit_bias(db_h0 + stash[2304 + 1024 + own_q * 64 + lane], own_q, j, g, pbias ? pbias + kBiasHead0 : nullptr,
                     grads.head_b0, 64);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    coop_emit(dW_h1[i], own_n, own_m2 + i, 64, prow ? prow + kOffHead1 : nullptr, grads.head_W1, 64, 64, false, 0, j, g);
    if (!RAYC)
      coop_emit(dW_h0[i], own_n, own_m2 + i, 64, prow ? prow + kOffHead0 : nullptr, grads.head_W0, 64, 31 + app_dim, true,
                app_dim, j, g);
  }
  if (RAYC) {  // the per-ray columns: slot tiles 0 (SH), 2, 3 (appearance); zeros where the field has no appearance embedding
    coop_emit(dW_r[0], wave & 3, wave < 4 ? 0 : 2, 64, prow ? prow + kOffHead0 : nullptr, grads.head_W0, 64, 31 + app_dim, true,
              app_dim, j, g);
    if (wave < 4)
      coop_emit(dW_r[1], wave, 3, 64, prow ? prow + kOffHead0 : nullptr, grads.head_W0, 64, 31 + app_dim, true, app_dim, j, g);
  }
  coop_emit(dW_b0[0], own_n, own_m1, 32, prow ? prow + kOffBase0 : nullptr, grads.base_W0, 64, 32, false, 0, j, g);
  if (bias_owner44) {
    coop_emit_bias(db_h1, own_n, j, g, pbias ? pbias + kBiasHead1 : nullptr, grads.head_b1, 64);
    if (!RAYC) coop_emit_bias(db_h0, own_n, j, g, pbias ? pbias + kBiasHead0 : nullptr, grads.head_b0, 64);
  }
  if (own_m1 == 0) coop_emit_bias(db_b0, own_n, j, g, pbias ? pbias + kBiasBase0 : nullptr, grads.base_b0, 64);
  PROBE_STAMP(kCoopWaves, 63);
}

__global__ __launch_bounds__(kReduceThreads) void field_dw_reduce_kernel(const float* __restrict__ partials,
                                                                          int num_partials,
                                                                          nsamd_field_mlp_grads grads, int app_dim,
                                                                          const float* __restrict__ app_rows,
                                                                          const int64_t* __restrict__ cams,
                                                                          int64_t num_rays, int tiles_per_ray) {
  extern __shared__ __attribute__((aligned(16))) float red_lds[];  // (field_reduce.h)
  field_dw_reduce_body(red_lds, (int)blockIdx.x, partials, num_partials, grads, app_dim, app_rows, cams, num_rays, tiles_per_ray);
}

// one MFMA with the assumed operand / result lane mapping (layout probe for the tests)
__global__ void probe_mfma16_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                    float* __restrict__ out) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = mfma16(A[j * 4 + g], B[g * 16 + j], c);  // A[16][4] row-major, B[4][16] row-major
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + j] = c[r];
}

}  // namespace nsamd

using namespace nsamd;

static int field_common_checks(const float* enc, const float* directions, int64_t dir_group, int64_t M,
                               const nsamd_field_mlp& mlp, const int64_t* cams, const float* app_const, int* app_dim) {
  NSAMD_REQUIRE(M >= 0 && dir_group >= 1);
  if (M > kMaxFieldPoints) return NSAMD_ERR_UNSUPPORTED;  // 32-bit per-lane offsets into the [32, M] feature matrix
  NSAMD_REQUIRE(enc && directions);
  NSAMD_REQUIRE(mlp.base_W0 && mlp.base_b0 && mlp.base_W1 && mlp.base_b1 && mlp.head_W0 && mlp.head_b0 &&
                mlp.head_W1 && mlp.head_b1 && mlp.head_W2 && mlp.head_b2);
  if (cams != nullptr) {
    NSAMD_REQUIRE(mlp.appearance != nullptr && mlp.num_images > 0);
    *app_dim = 32;
  } else if (app_const != nullptr) {
    *app_dim = 32;
  } else {
    *app_dim = 0;  // field built without an appearance embedding: head input is SH16 | geo15
  }
  return NSAMD_OK;
}

static int num_cus() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cached;
}

// Compute units the persistent backward leaves to other streams' launches (nsamd_field_mlp_bwd_reserve_cus): its workgroups
// own a CU's LDS and registers for the whole launch, so whatever is queued beside it otherwise waits for its end.
static int g_bwd_reserved_cus = 0;

// workgroups of a backward launch over `groups` tile groups (kCoopWaves tiles each). Reservation -1 = "one more sweep": the
// persistent workgroups take ceil(groups / workgroups) sweeps whatever the count, so a few CUs cannot be left out for free —
// the cheapest reservation is the one that adds exactly one sweep and spreads it evenly (1536 groups on 256 CUs: 6 sweeps of
// 256 -> 7 sweeps of 220, 36 CUs free); only where that costs <= 25 % (>= 4 sweeps).
static int field_bwd_workgroups(int64_t groups) {
  const int cus = num_cus();
  if (g_bwd_reserved_cus > 0) return cus - g_bwd_reserved_cus > 1 ? cus - g_bwd_reserved_cus : 1;
  if (g_bwd_reserved_cus < 0) {
    const int64_t sweeps = (groups + cus - 1) / cus;
    if (sweeps >= 4) return (int)((groups + sweeps) / (sweeps + 1));
  }
  return cus;
}

extern "C" int nsamd_field_mlp_bwd_reserve_cus(int cus) {
  const int prev = g_bwd_reserved_cus;
  g_bwd_reserved_cus = cus < 0 ? -1 : cus;
  return prev;
}

// the RAYC kernels apply: terms given and every 16-point tile inside one ray
static bool field_ray_terms_apply(const nsamd_field_mlp& mlp, int64_t dir_group, int64_t M) {
  static const bool on = getenv("NSAMD_RAY_TERMS") == nullptr || atoi(getenv("NSAMD_RAY_TERMS")) != 0;  // =0: A/B, the plain kernels
  return on && mlp.ray_terms != nullptr && dir_group % 16 == 0 && M % dir_group == 0;
}

extern "C" int nsamd_field_ray_terms(const float* directions, const int64_t* camera_indices, const float* appearance_const,
                                     int64_t num_rays, nsamd_field_mlp mlp, float* ray_terms, float* ray_inputs,
                                     nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(directions && ray_terms && mlp.head_W0 && mlp.head_b0);
  int app_dim = 0;
  if (camera_indices != nullptr) {
    NSAMD_REQUIRE(mlp.appearance != nullptr && mlp.num_images > 0);
    app_dim = 32;
  } else if (appearance_const != nullptr) {
    app_dim = 32;
  }
  const int64_t tiles = (num_rays + 15) / 16;
  const unsigned blocks = (unsigned)min((int64_t)num_cus(), (tiles + kWaves - 1) / kWaves);
  field_ray_terms_kernel<<<blocks, kFieldThreads, 0, (hipStream_t)stream>>>(directions, camera_indices, appearance_const, num_rays,
                                                                           mlp, app_dim, ray_terms, ray_inputs);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

static int field_mlp_fwd_impl(const float* enc, const float* selector, const float* directions,
                              const int64_t* camera_indices, const float* appearance_const, int64_t dir_group, int64_t M,
                              nsamd_field_mlp mlp, float* density, float* rgb, nsamd_stream_t stream) {
  if (M == 0) return NSAMD_OK;
  int app_dim = 0;
  int st = field_common_checks(enc, directions, dir_group, M, mlp, camera_indices, appearance_const, &app_dim);
  if (st) return st;
  NSAMD_REQUIRE(density != nullptr);  // rgb NULL: density only
  const size_t lds = sizeof(float) * (kFragTotal + 256);
  const int64_t tiles = (M + 15) / 16;
  // One 16-wave workgroup per CU by default (4 waves per SIMD, the weights staged once per CU): 57 us on the bench shape
  // against 59.5 (8 waves x 2 workgroups) and 65 (4 waves x 3) on the same box — NSAMD_FIELD_FWD_WAVES=8|4 selects those.
  static const int waves = getenv("NSAMD_FIELD_FWD_WAVES") ? atoi(getenv("NSAMD_FIELD_FWD_WAVES")) : 16;
  if (field_ray_terms_apply(mlp, dir_group, M) && rgb != nullptr) {
    const unsigned blocks = (unsigned)min((int64_t)num_cus(), (tiles + 15) / 16);
    field_mlp_fwd_kernel<16, true><<<blocks, 1024, lds, (hipStream_t)stream>>>(
        enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, density, rgb);
  } else if (waves == 16) {
    const unsigned blocks = (unsigned)min((int64_t)num_cus(), (tiles + 15) / 16);
    field_mlp_fwd_kernel<16><<<blocks, 1024, lds, (hipStream_t)stream>>>(
        enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, density, rgb);
  } else if (waves == 8) {
    const unsigned blocks = (unsigned)min((int64_t)num_cus() * 2, (tiles + 7) / 8);
    field_mlp_fwd_kernel<8><<<blocks, 512, lds, (hipStream_t)stream>>>(
        enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, density, rgb);
  } else {
    const unsigned blocks = (unsigned)min((int64_t)num_cus() * 3, (tiles + kWaves - 1) / kWaves);
    field_mlp_fwd_kernel<kWaves><<<blocks, kFieldThreads, lds, (hipStream_t)stream>>>(
        enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, density, rgb);
  }
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_field_mlp_fwd(const float* enc, const float* selector, const float* directions,
                                   const int64_t* camera_indices, const float* appearance_const, int64_t dir_group,
                                   int64_t M, nsamd_field_mlp mlp, float* density, float* rgb,
                                   nsamd_stream_t stream) {
  return field_mlp_fwd_impl(enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, density,
                            rgb, stream);
}

static int field_mlp_bwd_impl(const float* enc, const float* selector, const float* directions,
                              const int64_t* camera_indices, const float* appearance_const, int64_t dir_group, int64_t M,
                              nsamd_field_mlp mlp, const float* ddensity, const float* drgb, float* denc,
                              nsamd_field_mlp_grads grads, float* workspace, int64_t workspace_floats,
                              nsamd_stream_t stream, int phases = 3, const RouteArgs* route_in = nullptr,
                              float* dtable = nullptr, float* scatter_ws = nullptr, int64_t scatter_ws_floats = 0) {
  // phases: 1 = the gradient kernel (denc + per-workgroup partials), 2 = the fixed-order sum of the partials, 3 = both
  if (M == 0) return NSAMD_OK;
  int app_dim = 0;
  int st = field_common_checks(enc, directions, dir_group, M, mlp, camera_indices, appearance_const, &app_dim);
  if (st) return st;
  NSAMD_REQUIRE(ddensity && drgb && (denc || route_in));
  const int64_t tiles = (M + 15) / 16;
  const size_t lds = sizeof(float) * (kRowTotal + 256 + kCoopWaves * 2 * kScratchTile);
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NSAMD_ERR_NO_DEVICE;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {  // the dynamic-LDS opt-in is per device
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&field_mlp_bwd_kernel<false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&field_mlp_bwd_kernel<true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + sizeof(uint32_t) * kRouteLdsWords)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&field_mlp_bwd_kernel<false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&field_mlp_bwd_kernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + sizeof(uint32_t) * kRouteLdsWords)) != hipSuccess)
      return NSAMD_ERR_LAUNCH;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  static const int probe_skip = getenv("NSAMD_FIELD_BWD_SKIP") ? atoi(getenv("NSAMD_FIELD_BWD_SKIP")) : 0;
  const int64_t groups = (tiles + kCoopWaves - 1) / kCoopWaves;
  const unsigned blocks = (unsigned)min((int64_t)field_bwd_workgroups(groups), groups);
  float* partials = (workspace != nullptr && workspace_floats >= (int64_t)blocks * kPartialStride) ? workspace : nullptr;
  // per-tile rows of the appearance-embedding gradient (fixed-order reduction per camera): needs every 16-point tile
  // inside one ray and room behind the weight-gradient partials; otherwise float atomics (sums in no fixed order)
  float* app_partials = nullptr;
  int app_rows_per_point = 0;
  // ray terms (RAYC kernels): need the partial rows (their per-ray columns of head layer 0 are written, never added) and, for
  // the appearance rows, the per-tile scratch below
  if (partials != nullptr && camera_indices != nullptr && grads.appearance != nullptr && dir_group % 16 == 0 &&
      M % dir_group == 0 && mlp.num_images <= 8192 &&
      workspace_floats >= (int64_t)blocks * kPartialStride + tiles * 32) {
    app_partials = workspace + (int64_t)blocks * kPartialStride;
  } else if (partials != nullptr && camera_indices != nullptr && grads.appearance != nullptr && dir_group == 1 &&
             mlp.num_images <= 8192 && workspace_floats >= (int64_t)blocks * kPartialStride + M * 32) {
    app_partials = workspace + (int64_t)blocks * kPartialStride;  // one row per POINT (a camera index per sample)
    app_rows_per_point = 1;
  }
  const bool rayc = partials != nullptr && field_ray_terms_apply(mlp, dir_group, M) && mlp.ray_inputs != nullptr &&
                    (app_partials != nullptr || camera_indices == nullptr || grads.appearance == nullptr);
  if (phases != 3) NSAMD_REQUIRE(partials != nullptr);  // without scratch the kernel flushes with atomics: nothing to split
  ScatterPlan plan{};
  if (route_in != nullptr) {
    // producer mode: the kernel emits the scatter's pass-1 records (one static segment per workgroup and tile)
    NSAMD_REQUIRE(dtable != nullptr && scatter_ws != nullptr && partials != nullptr);
    plan = scatter_plan_producers(route_in->grid, M, (int)blocks, kProducerSegCap);
    if (!plan.ok) return NSAMD_ERR_UNSUPPORTED;
    NSAMD_REQUIRE(scatter_ws_floats >= plan.total_words);
    RouteArgs R = *route_in;
    R.G = plan.geom;
    R.buf = scatter_bufs(scatter_ws, plan);
    R.buf.log2_table_size = R.grid.log2_table_size;
    if (phases & 1) {
      if (rayc)
        field_mlp_bwd_kernel<true, true><<<blocks, kCoopThreads, lds + sizeof(uint32_t) * kRouteLdsWords, (hipStream_t)stream>>>(
            enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, ddensity, drgb, denc,
            grads, partials, app_partials, app_rows_per_point, probe_skip, R);
      else
        field_mlp_bwd_kernel<true, false><<<blocks, kCoopThreads, lds + sizeof(uint32_t) * kRouteLdsWords, (hipStream_t)stream>>>(
            enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, ddensity, drgb, denc,
            grads, partials, app_partials, app_rows_per_point, probe_skip, R);
      NSAMD_CHECK_LAUNCH();
    }
  } else if (phases & 1) {
    if (rayc)
      field_mlp_bwd_kernel<false, true><<<blocks, kCoopThreads, lds, (hipStream_t)stream>>>(
          enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, ddensity, drgb, denc,
          grads, partials, app_partials, app_rows_per_point, probe_skip, RouteArgs{});
    else
      field_mlp_bwd_kernel<false, false><<<blocks, kCoopThreads, lds, (hipStream_t)stream>>>(
          enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, app_dim, ddensity, drgb, denc,
          grads, partials, app_partials, app_rows_per_point, probe_skip, RouteArgs{});
    NSAMD_CHECK_LAUNCH();
  }
  // weight-gradient partials -> gradients, and (extra blocks, one per camera) the appearance rows -> embedding gradient
  const unsigned app_blocks = app_partials != nullptr ? (unsigned)mlp.num_images : 0u;
  const bool reduce = partials != nullptr && (phases & 2);
  const bool apply = route_in != nullptr && (phases & 4);
  // Both asked for in one call: the reduce RIDES the apply pass as extra workgroups (field_reduce.h; the two are independent)
  // — one launch and one dependent-launch gap fewer on the critical path, same sums in the same order.
  // NSAMD_REDUCE_RIDER=0: the reduce as a launch of its own (A/B).
  static const bool rider_on = getenv("NSAMD_REDUCE_RIDER") == nullptr || atoi(getenv("NSAMD_REDUCE_RIDER")) != 0;
  const bool ride = reduce && apply && rider_on && scatter_apply_takes_rider(plan);
  if (reduce && !ride) {
    const size_t red_lds = sizeof(float) * kReduceGroups * 64;
    field_dw_reduce_kernel<<<kDwBlocks + app_blocks, kReduceThreads, red_lds, (hipStream_t)stream>>>(
        partials, (int)blocks, grads, app_dim, app_partials, camera_indices, app_rows_per_point ? M : M / dir_group,
        app_rows_per_point ? 1 : (int)(dir_group / 16));
    NSAMD_CHECK_LAUNCH();
  }
  if (apply) {  // pass 2 over the records the kernel left in the queues: the table's gradient is WRITTEN
    ReduceRider rd{};
    if (ride) {
      rd.partials = partials, rd.num_partials = (int)blocks, rd.grads = grads, rd.app_dim = app_dim, rd.app_rows = app_partials;
      rd.cams = camera_indices, rd.num_rays = app_rows_per_point ? M : M / dir_group;
      rd.tiles_per_ray = app_rows_per_point ? 1 : (int)(dir_group / 16);
      rd.blocks = (int)(kDwBlocks + app_blocks);
    }
    return scatter_apply_launch(route_in->grid, plan, scatter_ws, dtable, /*overwrite=*/true, (hipStream_t)stream,
                                ride ? &rd : nullptr);
  }
  return NSAMD_OK;
}

static unsigned field_bwd_blocks(int64_t M) {
  const int64_t tiles = (M + 15) / 16;
  return (unsigned)min((int64_t)num_cus(), (tiles + kCoopWaves - 1) / kCoopWaves);
}

extern "C" int64_t nsamd_field_mlp_bwd_scatter_workspace(nsamd_grid grid, int64_t M, int64_t* state_words) {
  if (M <= 0 || grid.num_levels != 16) return 0;
  const ScatterPlan p = scatter_plan_producers(grid, M, (int)field_bwd_blocks(M), kProducerSegCap);
  if (!p.ok) return 0;
  if (state_words != nullptr) *state_words = p.state_words;
  return p.total_words;
}

extern "C" int nsamd_field_mlp_bwd_scatter_phase(nsamd_points pts, int transform, nsamd_aabb aabb, nsamd_grid grid,
                                                 const float* enc, const float* selector, const float* directions,
                                                 const int64_t* camera_indices, const float* appearance_const,
                                                 int64_t dir_group, int64_t M, nsamd_field_mlp mlp, const float* ddensity,
                                                 const float* drgb, float* denc, nsamd_field_mlp_grads grads,
                                                 float* workspace, int64_t workspace_floats, float* dtable,
                                                 float* scatter_workspace, int64_t scatter_workspace_floats, int phase,
                                                 nsamd_stream_t stream) {
  NSAMD_REQUIRE(phase == 1 || phase == 2 || phase == 4 || phase == 6 || phase == 7);
  if (M == 0) return NSAMD_OK;
  if (grid.num_levels != 16) return NSAMD_ERR_UNSUPPORTED;  // 32 features = the K of base layer 0
  NSAMD_REQUIRE(M > 0 && transform >= 0 && transform <= 2 && grid.log2_table_size >= 1 && grid.log2_table_size <= 28);
  if (pts.positions == nullptr) {
    NSAMD_REQUIRE(pts.origins && pts.directions && pts.t_bins && pts.samples_per_ray > 0 && M % pts.samples_per_ray == 0);
  }
  NSAMD_REQUIRE(dtable != nullptr && scatter_workspace != nullptr && workspace != nullptr);
  RouteArgs R{};
  R.P = pts;
  R.transform = transform;
  R.box = aabb;
  R.grid = grid;
  return field_mlp_bwd_impl(enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, ddensity, drgb,
                            denc, grads, workspace, workspace_floats, stream, phase, &R, dtable, scatter_workspace,
                            scatter_workspace_floats);
}

extern "C" int nsamd_field_mlp_bwd_scatter(nsamd_points pts, int transform, nsamd_aabb aabb, nsamd_grid grid,
                                           const float* enc, const float* selector, const float* directions,
                                           const int64_t* camera_indices, const float* appearance_const,
                                           int64_t dir_group, int64_t M, nsamd_field_mlp mlp, const float* ddensity,
                                           const float* drgb, float* denc, nsamd_field_mlp_grads grads, float* workspace,
                                           int64_t workspace_floats, float* dtable, float* scatter_workspace,
                                           int64_t scatter_workspace_floats, nsamd_stream_t stream) {
  return nsamd_field_mlp_bwd_scatter_phase(pts, transform, aabb, grid, enc, selector, directions, camera_indices,
                                           appearance_const, dir_group, M, mlp, ddensity, drgb, denc, grads, workspace,
                                           workspace_floats, dtable, scatter_workspace, scatter_workspace_floats, 7, stream);
}

extern "C" int nsamd_field_mlp_bwd(const float* enc, const float* selector, const float* directions,
                                   const int64_t* camera_indices, const float* appearance_const, int64_t dir_group,
                                   int64_t M, nsamd_field_mlp mlp, const float* ddensity, const float* drgb,
                                   float* denc, nsamd_field_mlp_grads grads, float* workspace,
                                   int64_t workspace_floats, nsamd_stream_t stream) {
  return field_mlp_bwd_impl(enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, ddensity,
                            drgb, denc, grads, workspace, workspace_floats, stream);
}

extern "C" int nsamd_field_mlp_bwd_phase(const float* enc, const float* selector, const float* directions,
                                         const int64_t* camera_indices, const float* appearance_const,
                                         int64_t dir_group, int64_t M, nsamd_field_mlp mlp, const float* ddensity,
                                         const float* drgb, float* denc, nsamd_field_mlp_grads grads, float* workspace,
                                         int64_t workspace_floats, int phase, nsamd_stream_t stream) {
  NSAMD_REQUIRE(phase == 1 || phase == 2);
  return field_mlp_bwd_impl(enc, selector, directions, camera_indices, appearance_const, dir_group, M, mlp, ddensity,
                            drgb, denc, grads, workspace, workspace_floats, stream, phase);
}

extern "C" int nsamd_probe_mfma_bf16(const float* A, const float* B, float* out, nsamd_stream_t stream) {
  NSAMD_REQUIRE(A && B && out);
  probe_mfma_bf16_kernel<<<1, 64, 0, (hipStream_t)stream>>>(A, B, out);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_probe_mfma16(const float* A, const float* B, float* out, nsamd_stream_t stream) {
  NSAMD_REQUIRE(A && B && out);
  probe_mfma16_kernel<<<1, 64, 0, (hipStream_t)stream>>>(A, B, out);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
