// GPU diagnostic (not part of the product): throughput of LDS atomics on gfx950.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/probe_lds_atomics.hip -o /tmp/probe_lds && /tmp/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>  // 0: ds_add_f32 random, 1: ds_add_u32 random, 2: plain RMW random (no atomic), 3: ds_add_f32 same address/wave
__global__ __launch_bounds__(1024) void k(float* out, int iters, int entries) {
  extern __shared__ float acc[];
  for (int e = threadIdx.x; e < entries; e += 1024) acc[e] = 0.f;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    uint32_t a = (MODE == 3) ? ((s >> 8) & ~63u) % entries : (s >> 8) % entries;
    if (MODE == 3) a = __builtin_amdgcn_readfirstlane(a);
    if (MODE == 0 || MODE == 3) atomicAdd(acc + a, 1.0f);
    else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(acc) + a, 1u);
    else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (a >> 1), (unsigned long long)(s | 1u) << 7);
    else if (MODE == 5) {  // float add through an integer compare-and-swap loop
      uint32_t* w = reinterpret_cast<uint32_t*>(acc) + a;
      uint32_t old = *w, assumed;
      do {
        assumed = old;
        old = atomicCAS(w, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f));
      } while (old != assumed);
    } else acc[a] += 1.0f;
  }
  __syncthreads();
  float t = 0.f;
  for (int e = threadIdx.x; e < entries; e += 1024) t += acc[e];
  if (t == -1.f) out[blockIdx.x] = t;
}
template <int MODE> void run(const char* name, int entries) {
  float* out; hipMalloc(&out, 4096);
  const int iters = 512, blocks = 512;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 1024, entries * 4>>>(out, iters, entries);
  hipEventRecord(a);
  k<MODE><<<blocks, 1024, entries * 4>>>(out, iters, entries);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 1024 * iters;
  printf("%-34s entries=%6d : %8.1f us  %8.1f G lane-ops/s  (%.2f lane-ops/clk/CU @2.4GHz,256CU)\n", name, entries, ms * 1e3,
         ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
  hipFree(out);
}
int main() {
  for (int entries : {32768, 2048}) {
    run<0>("ds_add_f32 random", entries);
    run<1>("ds_add_u32 random", entries);
    run<2>("plain LDS RMW random (no atomic)", entries);
    run<3>("ds_add_f32 one address per wave", entries);
    run<4>("ds_add_u64 random", entries);
    run<5>("float add via ds CAS loop, random", entries);
  }
  return 0;
}
