// Generic dense layer  y = act(x W^T + b)  on the gfx950 f32 MFMA, for the stand-alone `MLP` of the plugin API
// (reference: MLP.pytorch_fwd, /root/reference/nerfstudio/field_components/mlp.py:160-179, arbitrary widths).
// The nerfacto shapes never come here — they run in the fused field kernels (field_mlp.hip, density_mlp.hip); this is
// the general-shape path with the same building blocks: 16-point tiles per wavefront, chain-layout operands
// (lane (j = lane&15, g = lane>>4) holds feature 16t+4g+r of point j), weights staged once per workgroup in LDS as
// MFMA fragments. Row-major x / y: a lane's 4 features are one 16-B access.
//   forward    y  = act(x W^T + b)
//   data grad  dx = dpre W,            dpre = dy * act'(y)   (same kernel, transposed fragments)
//   weight grad dW += dpre^T x, db += sum dpre   (one wave per 16x16 dW tile and chunk of points; the A/B operands of
//              this reduction-over-points GEMM are contiguous 64-B row segments of dpre / x, so no transposes)
// Layers wider than 128 (vanilla-nerf's 8 x 256 MLP with its 319-wide skip layer, mlp.py:143-158) run as a grid of
// 128 x 128 blocks of W: one launch per block, the partial sums of a row of blocks accumulate in the output and the
// activation is applied by the last block.
#include "common.h"

namespace nsamd {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kLinThreads = 256;
constexpr int kLinWaves = 4;

__device__ __forceinline__ v4f mfma16l(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float act_grad(int act, float y) {
  if (act == 1) return y > 0.0f ? 1.0f : 0.0f;   // ReLU (through the post-activation value)
  if (act == 2) return y * (1.0f - y);            // Sigmoid
  if (act == 3) return 1.0f - expf(-y);           // Softplus: sigmoid(x) = 1 - exp(-softplus(x))
  return 1.0f;
}

// 4 consecutive features [c0, c0+4) of row p of a row-major matrix with row stride ld, zero beyond `cols`
__device__ __forceinline__ v4f load_row4(const float* __restrict__ x, int64_t p, int ld, int cols, int c0, bool vec) {
  v4f v = {0.f, 0.f, 0.f, 0.f};
  const float* row = x + p * (int64_t)ld;
  if (vec && c0 + 3 < cols) {
    v = *reinterpret_cast<const v4f*>(row + c0);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c0 + r < cols) v[r] = row[c0 + r];
  }
  return v;
}

// One block of a dense layer: `in_cols` input features starting at `in`, `out_cols` outputs starting at `out` (both already
// offset to the block, row strides in_ld / out_ld); Wb = the block's corner of W, row stride w_ld.
struct LinBlock {
  const float* in; int in_ld; int in_cols;
  const float* W; int w_ld;
  const float* bias;          // forward, first block of a row only
  const float* y_for_grad; int y_ld;  // data gradient: post-activation values of the INPUT side (dy's layer output)
  float* out; int out_ld; int out_cols;
  int act;                    // forward: applied to the finished sum (last block of a row); data gradient: act'(y) factor
  int accumulate;             // start from the partial sums already in `out`
};

// TRANSPOSED = false: frag[n][t][lane][r] = Wb[16n + j][16t + 4g + r]      (y = x W^T;  W is [N, K])
// TRANSPOSED = true : frag[n][t][lane][r] = Wb[16t + 4g + r][16n + j]      (dx = dpre W: output tile n over K, input t over N)
template <int NT, int KT, bool TRANSPOSED>
__global__ __launch_bounds__(kLinThreads) void linear_chain_kernel(LinBlock B, int64_t M) {
  // forward: in = x, out = y.   data grad: in = dy, y_for_grad = y, out = dx.
  extern __shared__ __attribute__((aligned(16))) float frag[];
  for (int e = threadIdx.x; e < NT * KT * 256; e += kLinThreads) {
    const int r = e & 3, lane = (e >> 2) & 63, tile = e >> 8;
    const int t = tile % KT, n = tile / KT;
    const int j = lane & 15, g = lane >> 4;
    const int o = 16 * n + j, i = 16 * t + 4 * g + r;  // output index / input index of this element
    float v = 0.0f;
    if (o < B.out_cols && i < B.in_cols) v = TRANSPOSED ? B.W[(int64_t)i * B.w_ld + o] : B.W[(int64_t)o * B.w_ld + i];
    frag[e] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int64_t tiles = (M + 15) / 16;
  const bool in_vec = (B.in_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(B.in) & 15) == 0;
  const bool y_vec = B.y_for_grad != nullptr && (B.y_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(B.y_for_grad) & 15) == 0;
  const bool out_vec = (B.out_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(B.out) & 15) == 0;
  for (int64_t tile = (int64_t)blockIdx.x * kLinWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kLinWaves) {
    asm volatile("" ::: "memory");  // keep the (loop-invariant) fragments in LDS, not hoisted into 100s of VGPRs
    const int64_t p = tile * 16 + j;
    const bool live = p < M;
    const int64_t pc = live ? p : M - 1;
    v4f x[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      x[t] = load_row4(B.in, pc, B.in_ld, B.in_cols, 16 * t + 4 * g, in_vec);
      if (TRANSPOSED) {  // dpre = dy * act'(y)
        const v4f yv = (B.act != 0) ? load_row4(B.y_for_grad, pc, B.y_ld, B.in_cols, 16 * t + 4 * g, y_vec)
                                    : v4f{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] *= act_grad(B.act, yv[r]);
      }
    }
    v4f acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (B.accumulate) {
        acc[n] = load_row4(B.out, pc, B.out_ld, B.out_cols, 16 * n + 4 * g, out_vec);
      } else {
        acc[n] = v4f{0.f, 0.f, 0.f, 0.f};
        if (!TRANSPOSED && B.bias != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[n][r] = (16 * n + 4 * g + r < B.out_cols) ? B.bias[16 * n + 4 * g + r] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      v4f a[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) a[n] = *reinterpret_cast<const v4f*>(frag + ((n * KT + t) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = mfma16l(a[n][r], x[t][r], acc[n]);
    }
    if (live) {
      float* orow = B.out + p * (int64_t)B.out_ld;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[n][r];
          if (!TRANSPOSED) {
            if (B.act == 1) v = fmaxf(v, 0.0f);
            else if (B.act == 2) v = 1.0f / (1.0f + expf(-v));
            else if (B.act == 3) v = v > 20.0f ? v : log1pf(expf(v));  // torch.nn.Softplus(beta = 1, threshold = 20)
          }
          acc[n][r] = v;
        }
        const int c0 = 16 * n + 4 * g;
        if (out_vec && c0 + 3 < B.out_cols) {
          *reinterpret_cast<v4f*>(orow + c0) = acc[n];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + r < B.out_cols) orow[c0 + r] = acc[n][r];
        }
      }
    }
  }
}

// dW[16n.., 16m..] += sum_p dpre[p][.] x[p][.]; one wave per (tile, chunk); db from the m == 0 tiles
__global__ __launch_bounds__(kLinThreads) void linear_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ dy, int64_t M, int K, int N,
                                                               int act, int KT, int chunks, float* __restrict__ dW,
                                                               float* __restrict__ db) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int64_t unit = (int64_t)blockIdx.x * kLinWaves + wave;  // = tile * chunks + chunk
  const int tile = (int)(unit / chunks), chunk = (int)(unit % chunks);
  const int n = tile / KT, m = tile % KT;
  if (16 * n >= N) return;
  const int64_t per = ((M + chunks - 1) / chunks + 3) & ~(int64_t)3;
  const int64_t p0 = (int64_t)chunk * per, p1 = min(M, p0 + per);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.0f;
  const int o = 16 * n + j, i = 16 * m + j;
  for (int64_t q = p0; q < p1; q += 4) {  // wave-uniform trip count; MFMA step = points q + g, g = 0..3
    const int64_t p = q + g;
    float a = 0.0f, b = 0.0f;
    if (p < p1) {
      if (o < N) {
        a = dy[p * N + o];
        if (act != 0) a *= act_grad(act, y[p * N + o]);
      }
      if (i < K) b = x[p * K + i];
    }
    bsum += a;
    acc = mfma16l(a, b, acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * n + 4 * g + r, col = 16 * m + j;
    if (row < N && col < K) unsafeAtomicAdd(dW + (int64_t)row * K + col, acc[r]);
  }
  if (db != nullptr && m == 0) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (g == 0 && o < N) unsafeAtomicAdd(db + o, bsum);
  }
}

static int pad_tiles(int dim) {  // tiles of 16, rounded up to 1, 2, 4, 8 (widths up to 128)
  const int t = (dim + 15) / 16;
  int p = 1;
  while (p < t) p <<= 1;
  return p;
}

template <bool TR>
static int launch_block(const LinBlock& B, int64_t M, hipStream_t st) {
  const int NT = pad_tiles(B.out_cols), KT = pad_tiles(B.in_cols);
  const size_t lds = sizeof(float) * (size_t)NT * KT * 256;
  if (NT > 8 || KT > 8) return NSAMD_ERR_UNSUPPORTED;  // blocks of up to 128 x 128
  const int64_t tiles = (M + 15) / 16;
  const unsigned blocks = (unsigned)min((int64_t)1024, (tiles + kLinWaves - 1) / kLinWaves);
#define NSAMD_LIN_CASE(nt, kt)                                                                                          \
  if (NT == nt && KT == kt) {                                                                                          \
    if (lds > 64 * 1024)                                                                                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_chain_kernel<nt, kt, TR>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    linear_chain_kernel<nt, kt, TR><<<blocks, kLinThreads, lds, st>>>(B, M);                                           \
    NSAMD_CHECK_LAUNCH();                                                                                              \
    return NSAMD_OK;                                                                                                   \
  }
#define NSAMD_LIN_ROW(nt) NSAMD_LIN_CASE(nt, 1) NSAMD_LIN_CASE(nt, 2) NSAMD_LIN_CASE(nt, 4) NSAMD_LIN_CASE(nt, 8)
  NSAMD_LIN_ROW(1) NSAMD_LIN_ROW(2) NSAMD_LIN_ROW(4) NSAMD_LIN_ROW(8)
#undef NSAMD_LIN_ROW
#undef NSAMD_LIN_CASE
  return NSAMD_ERR_UNSUPPORTED;
}

constexpr int kLinBlock = 128;

// y = act(x W^T + b): for every 128-wide block of outputs, the 128-wide blocks of inputs one after the other
static int linear_forward(const float* x, const float* W, const float* b, int64_t M, int K, int N, int act, float* y,
                          hipStream_t st) {
  for (int n0 = 0; n0 < N; n0 += kLinBlock) {
    for (int k0 = 0; k0 < K; k0 += kLinBlock) {
      LinBlock B{};
      B.in = x + k0; B.in_ld = K; B.in_cols = min(kLinBlock, K - k0);
      B.W = W + (int64_t)n0 * K + k0; B.w_ld = K;
      B.bias = (k0 == 0 && b != nullptr) ? b + n0 : nullptr;
      B.out = y + n0; B.out_ld = N; B.out_cols = min(kLinBlock, N - n0);
      B.act = (k0 + kLinBlock >= K) ? act : 0;
      B.accumulate = k0 > 0;
      const int s = launch_block<false>(B, M, st);
      if (s) return s;
    }
  }
  return NSAMD_OK;
}

// dx = (dy * act'(y)) W: for every 128-wide block of inputs of the layer (= outputs here), the blocks of neurons in turn
static int linear_data_grad(const float* W, const float* y, const float* dy, int64_t M, int K, int N, int act, float* dx,
                            hipStream_t st) {
  for (int k0 = 0; k0 < K; k0 += kLinBlock) {
    for (int n0 = 0; n0 < N; n0 += kLinBlock) {
      LinBlock B{};
      B.in = dy + n0; B.in_ld = N; B.in_cols = min(kLinBlock, N - n0);
      B.W = W + (int64_t)n0 * K + k0; B.w_ld = K;
      B.y_for_grad = y != nullptr ? y + n0 : nullptr; B.y_ld = N;
      B.out = dx + k0; B.out_ld = K; B.out_cols = min(kLinBlock, K - k0);
      B.act = act;
      B.accumulate = n0 > 0;
      const int s = launch_block<true>(B, M, st);
      if (s) return s;
    }
  }
  return NSAMD_OK;
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_linear_fwd(const float* x, const float* W, const float* b, int64_t M, int32_t K, int32_t N,
                                int activation, float* y, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && K > 0 && N > 0 && activation >= 0 && activation <= 3);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(x && W && y);
  return linear_forward(x, W, b, M, K, N, activation, y, (hipStream_t)stream);
}

extern "C" int nsamd_linear_bwd(const float* x, const float* W, const float* y, const float* dy, int64_t M, int32_t K,
                                int32_t N, int activation, float* dx, float* dW, float* db, nsamd_stream_t stream) {
  NSAMD_REQUIRE(M >= 0 && K > 0 && N > 0 && activation >= 0 && activation <= 3);
  if (M == 0) return NSAMD_OK;
  NSAMD_REQUIRE(x && W && dy && (activation == 0 || y != nullptr));
  hipStream_t st = (hipStream_t)stream;
  if (dx != nullptr) {
    const int s = linear_data_grad(W, y, dy, M, K, N, activation, dx, st);
    if (s) return s;
  }
  if (dW != nullptr) {
    const int NT = (N + 15) / 16, KT = (K + 15) / 16;
    int chunks = (int)min((int64_t)256, max((int64_t)1, M / 1024));
    while ((int64_t)NT * KT * chunks > 16384 && chunks > 1) chunks >>= 1;
    const int64_t units = (int64_t)NT * KT * chunks;
    linear_dw_kernel<<<(unsigned)((units + kLinWaves - 1) / kLinWaves), kLinThreads, 0, st>>>(x, y, dy, M, K, N, activation,
                                                                                           KT, chunks, dW, db);
    NSAMD_CHECK_LAUNCH();
  }
  return NSAMD_OK;
}
