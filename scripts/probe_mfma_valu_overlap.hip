// Does the f32-input MFMA (v_mfma_f32_16x16x4_f32) run beside VALU work of the SIMD's other wave, the way the bf16 MFMA does?
// One workgroup per CU; waves 0-3 (one per SIMD) issue matrix instructions, waves 4-7 issue v_fma_f32 chains. Timed: matrix waves
// alone, vector waves alone, both together. Together ~ max(alone) = separate pipes; together ~ sum = one shared datapath.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_mfma_valu_overlap.hip -o /tmp/probe_overlap && /tmp/probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>  // 0: f32 MFMA 16x16x4, 1: bf16 MFMA 16x16x32
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {  // mode bit 0: matrix waves work, bit 1: vector waves work
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc_out = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      v4f c[8];
      for (int i = 0; i < 8; ++i) c[i] = v4f{0.f, 0.f, 0.f, 0.f};
      float a = 1.0f + lane * 1e-3f, b = 0.5f;
      bf16x8 ah, bh;
      for (int q = 0; q < 8; ++q) { ah[q] = (__bf16)(1.0f + q); bh[q] = (__bf16)(0.25f * lane); }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
          else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c[i], 0, 0, 0);
        }
      }
      for (int i = 0; i < 8; ++i) acc_out += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    }
  } else if (mode & 2) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + i + lane * 1e-3f;
    const float m = 0.999f, d = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], m, d);
    }
    for (int i = 0; i < 8; ++i) acc_out += x[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc_out;
}

template <int KIND>
static float run(float* out, int iters, int mode) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<256, 512>>>(out, iters, mode);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k<KIND><<<256, 512>>>(out, iters, mode);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;  // 32 000 matrix instructions per matrix wave; 256 000 v_fma per vector wave
  for (int kind = 0; kind < 2; ++kind) {
    float m = kind ? run<1>(out, iters, 1) : run<0>(out, iters, 1);
    float v = kind ? run<1>(out, iters, 2) : run<0>(out, iters, 2);
    float both = kind ? run<1>(out, iters, 3) : run<0>(out, iters, 3);
    printf("%s: matrix waves alone %.1f us (%.1f clk/instr at 2.4 GHz), vector waves alone %.1f us (%.2f clk/v_fma), together %.1f us  "
           "(max %.1f, sum %.1f)\n", kind ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32 ", m, m * 2400.f / (iters * 8.f), v,
           v * 2400.f / (iters * 64.f), both, m > v ? m : v, m + v);
  }
  return 0;
}
