// The main field's weight-gradient reduce (the fixed-order sum of the backward's per-workgroup partial rows, and the
// appearance-embedding gradient from its per-tile rows) as a device body: launched on its own by field_mlp.hip
// (field_dw_reduce_kernel) and as a RIDER of the table scatter's apply pass (scatter.hip: extra workgroups of that launch —
// the two need nothing of each other, and a launch of their own costs the reduce its ~12 us plus a dependent-launch gap).
// Not part of the C ABI.
#pragma once

#include "common.h"

namespace nsamd {

typedef float v4f __attribute__((ext_vector_type(4)));

// fragment sizes in floats: (N_out padded to 16) x (K padded to 16)
constexpr int kFragBase0 = 64 * 32, kFragBase1 = 16 * 64, kFragHead0 = 64 * 64, kFragHead1 = 64 * 64,
              kFragHead2 = 16 * 64;
constexpr int kOffBase0 = 0, kOffBase1 = kOffBase0 + kFragBase0, kOffHead0 = kOffBase1 + kFragBase1,
              kOffHead1 = kOffHead0 + kFragHead0, kOffHead2 = kOffHead1 + kFragHead1,
              kFragTotal = kOffHead2 + kFragHead2;  // 12288 floats = 48 KiB
constexpr int kBiasTotal = 64 + 16 + 64 + 64 + 16;  // padded biases
constexpr int kBiasBase0 = 0, kBiasBase1 = 64, kBiasHead0 = 80, kBiasHead1 = 144, kBiasHead2 = 208;
constexpr int kPartialStride = kFragTotal + 256;    // floats per workgroup in the weight-gradient partial buffer


// logical column of head layer 0 for internal slot s (-1: no column)
__device__ __forceinline__ int head0_col(int s, int app_dim) {
  if (s < 16) return s;
  if (s == 16) return -1;
  if (s < 32) return s - 1;
  return (s - 32 < app_dim) ? s - 1 : -1;
}


// destination of element e of the [kPartialStride] reduction layout (weights: padded [rows][slots] per layer, then
// the padded biases); nullptr for padding / absent tensors
__device__ __forceinline__ float* dw_destination(int e, const nsamd_field_mlp_grads& g, int app_dim) {
  auto weight = [&](float* base, int off, int n_real, int k_real, int k_pad, bool head0) -> float* {
    const int row = (e - off) / k_pad, slot = (e - off) - row * k_pad;
    const int col = head0 ? head0_col(slot, app_dim) : slot;
    return (base != nullptr && row < n_real && col >= 0 && col < k_real) ? base + row * k_real + col : nullptr;
  };
  if (e < kOffBase1) return weight(g.base_W0, kOffBase0, 64, 32, 32, false);
  if (e < kOffHead0) return weight(g.base_W1, kOffBase1, 16, 64, 64, false);
  if (e < kOffHead1) return weight(g.head_W0, kOffHead0, 64, 31 + app_dim, 64, true);
  if (e < kOffHead2) return weight(g.head_W1, kOffHead1, 64, 64, 64, false);
  if (e < kFragTotal) return weight(g.head_W2, kOffHead2, 3, 64, 64, false);
  const int b = e - kFragTotal;
  auto bias = [&](float* base, int off, int n_real) -> float* {
    return (base != nullptr && b - off < n_real) ? base + (b - off) : nullptr;
  };
  if (b < kBiasBase1) return bias(g.base_b0, kBiasBase0, 64);
  if (b < kBiasHead0) return bias(g.base_b1, kBiasBase1, 16);
  if (b < kBiasHead1) return bias(g.head_b0, kBiasHead0, 64);
  if (b < kBiasHead2) return bias(g.head_b1, kBiasHead1, 64);
  if (b < kBiasTotal) return bias(g.head_b2, kBiasHead2, 3);
  return nullptr;
}

// grads[...] += sum over workgroups of their partial weight gradients. 64 elements x 16 partial-groups per workgroup:
// every thread has its <= 16 loads in flight at once (the 12.8 MB of partials are a pure latency problem: the first
// version walked 64 partials per thread two at a time and took 15 us); the 16 group sums meet in LDS and one thread per
// element does the single-writer update.
constexpr int kReduceGroups = 16;
constexpr int kReduceThreads = 64 * kReduceGroups;
constexpr int kDwBlocks = (kPartialStride + 63) / 64;

// Appearance-embedding gradient from the per-tile rows of the backward (blocks >= kDwBlocks of the reduce launch, one
// per camera): thread t takes rays t, t + 1024, ... — their camera indices are fetched first, all in flight — and adds
// the rows of the rays that belong to this camera in ray order; the partial sums are folded by a fixed butterfly per wave and the 16 wave sums are
// added in wave order. Fixed assignment, fixed order: bit-reproducible (the float atomics this replaces were not).
__device__ void app_reduce_block(const float* __restrict__ rows, const int64_t* __restrict__ cams, int64_t num_rays,
                                 int tiles_per_ray, float* __restrict__ grad, int64_t cam, float* lds_part) {
  float acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) acc[k] = 0.0f;
  for (int64_t r0 = threadIdx.x; r0 < num_rays; r0 += (int64_t)kReduceThreads * 8) {
    bool mine[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t r = r0 + (int64_t)u * kReduceThreads;
      // (unconditional load, clamped row: predicated loads are waited for one by one — see the partial rows below)
      const int64_t c = cams[r < num_rays ? r : num_rays - 1];
      mine[u] = r < num_rays && c == cam;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!mine[u]) continue;
      const int64_t r = r0 + (int64_t)u * kReduceThreads;
      for (int t = 0; t < tiles_per_ray; ++t) {
        const v4f* row = reinterpret_cast<const v4f*>(rows + (r * tiles_per_ray + t) * 32);
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          const v4f v = row[k4];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[4 * k4 + c] += v[c];
        }
      }
    }
  }
  // wave-level butterfly (fixed tree), then the 16 wave sums per feature in wave order
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    float v = acc[k];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    acc[k] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 32; ++k) lds_part[wave * 32 + k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float tot = 0.0f;
    for (int w = 0; w < kReduceGroups; ++w) tot += lds_part[w * 32 + threadIdx.x];
    if (tot != 0.0f) grad[cam * 32 + threadIdx.x] += tot;
  }
}

__device__ __forceinline__ void field_dw_reduce_body(float* red_lds, int block, const float* __restrict__ partials,
                                                                          int num_partials,
                                                                          nsamd_field_mlp_grads grads, int app_dim,
                                                                          const float* __restrict__ app_rows,
                                                                          const int64_t* __restrict__ cams,
                                                                          int64_t num_rays, int tiles_per_ray) {
  if (block >= kDwBlocks) {
    app_reduce_block(app_rows, cams, num_rays, tiles_per_ray, grads.appearance, (int64_t)block - kDwBlocks, red_lds);
    return;
  }
  float(*part)[64] = reinterpret_cast<float(*)[64]>(red_lds);
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = block * 64 + el;
  float s = 0.f;
  if (e < kPartialStride) {
    for (int b0 = grp; b0 < num_partials; b0 += kReduceGroups * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        // UNCONDITIONAL loads (a row past the end re-reads the last one and is dropped below): written as
        // `b < n ? load : 0` every load sits in its own branch and the compiler waits for each with vmcnt(0) before it
        // issues the next — sixteen memory latencies in a row instead of one (read off the ISA; the launch took 15 us
        // for 12.8 MB)
        const int b = b0 + u * kReduceGroups;
        v[u] = partials[(size_t)(b < num_partials ? b : num_partials - 1) * kPartialStride + e];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (b0 + u * kReduceGroups < num_partials) ? v[u] : 0.0f;
    }
  }
  part[grp][el] = s;
  __syncthreads();
  if (grp == 0 && e < kPartialStride) {
    float* dst = dw_destination(e, grads, app_dim);
    if (dst != nullptr) {
      float t = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < kReduceGroups; ++g2) t += part[g2][el];
      *dst += t;
    }
  }
}


// What the apply pass carries along (scatter.hip): `blocks` extra workgroups of kReduceThreads threads run
// field_dw_reduce_body(block = 0 .. blocks - 1); blocks == 0: no rider.
struct ReduceRider {
  const float* partials;
  int num_partials;
  nsamd_field_mlp_grads grads;
  int app_dim;
  const float* app_rows;
  const int64_t* cams;
  int64_t num_rays;
  int tiles_per_ray;
  int blocks;
};

}  // namespace nsamd
