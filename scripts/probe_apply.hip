// GPU diagnostic (not part of the product): where does the tile-apply pass spend its time?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ void lds_add_pair(float* pair, float v0, float v1) {
  unsigned long long* w = reinterpret_cast<unsigned long long*>(pair);
  const unsigned long long old = *w;
  const float n0 = __uint_as_float((uint32_t)old) + v0, n1 = __uint_as_float((uint32_t)(old >> 32)) + v1;
  const unsigned long long want = (unsigned long long)__float_as_uint(n0) | ((unsigned long long)__float_as_uint(n1) << 32);
  if (atomicCAS(w, old, want) != old) { atomicAdd(pair, v0); atomicAdd(pair + 1, v1); }
}
// MODE 0: full; 1: no LDS accumulate (register sum); 2: no queue read (synthetic records);
// MODE 3: one private copy of the tile per wavefront (no cross-wave contention), summed at the end
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void apply(const uint4* __restrict__ queues, uint32_t n, uint32_t cap, int slice_log2,
                                                float* __restrict__ dtable, long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int entries = 1 << slice_log2;
  const int copies = (MODE == 3) ? THREADS / 64 : 1;
  float* mine = acc + (MODE == 3 ? (threadIdx.x >> 6) * 2 * entries : 0);
  long long t0 = clock64();
  for (int e = threadIdx.x; e < 2 * entries * copies; e += THREADS) acc[e] = 0.0f;
  __syncthreads();
  long long t1 = clock64();
  const uint4* q = queues + (size_t)blockIdx.x * cap;
  float dummy = 0.f;
  constexpr int kU = 8;
  uint32_t e = threadIdx.x;
  for (; e + (kU - 1) * THREADS < n; e += kU * THREADS) {
    uint4 r[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (MODE == 2) { uint32_t s = (e + u * THREADS) * 2654435761u; r[u] = make_uint4((s >> 9) & (entries - 1), 0x3f800000u, 0x3f800000u, 0); }
      else r[u] = q[e + u * THREADS];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (MODE == 1) dummy += __uint_as_float(r[u].y) + (float)r[u].x;
      else lds_add_pair(mine + 2 * r[u].x, __uint_as_float(r[u].y), __uint_as_float(r[u].z));
    }
  }
  __syncthreads();
  long long t2 = clock64();
  float4* out = reinterpret_cast<float4*>(dtable + ((size_t)blockIdx.x << slice_log2) * 2);
  const float4* a4 = reinterpret_cast<const float4*>(acc);
  for (int i = threadIdx.x; i < entries / 2; i += THREADS) {
    float4 o = out[i]; float4 a = a4[i];
    for (int c = 1; c < copies; ++c) { const float4 b = a4[i + c * (entries / 2)]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    o.x += a.x + dummy; o.y += a.y; o.z += a.z; o.w += a.w;
    out[i] = o;
  }
  __syncthreads();
  long long t3 = clock64();
  if (threadIdx.x == 0 && blockIdx.x < 64) { stamps[blockIdx.x * 4 + 0] = t1 - t0; stamps[blockIdx.x * 4 + 1] = t2 - t1; stamps[blockIdx.x * 4 + 2] = t3 - t2; }
}
static float g_hot_frac = 0.f; static int g_hot_n = 1;
template <int MODE, int THREADS> void run(const char* name, int tiles, int slice_log2, uint32_t n) {
  const uint32_t cap = n + 64;
  uint4* q; float* dt; long long* st;
  hipMalloc(&q, (size_t)tiles * cap * 16); hipMalloc(&dt, ((size_t)tiles << slice_log2) * 8); hipMalloc(&st, 64 * 4 * 8);
  hipMemset(dt, 0, ((size_t)tiles << slice_log2) * 8);
  std::vector<uint4> h((size_t)cap);
  uint32_t s = 12345;
  for (uint32_t i = 0; i < cap; ++i) {
    s = s * 1664525u + 1013904223u;
    uint32_t idx = (s >> 8) & ((1u << slice_log2) - 1);
    s = s * 1664525u + 1013904223u;
    if ((s >> 8) / 16777216.0f < g_hot_frac) idx = (idx % g_hot_n) * 37 % (1u << slice_log2);
    h[i] = make_uint4(idx, 0x3f800000u, 0x3f000000u, 0);
  }
  for (int t = 0; t < tiles; ++t) hipMemcpy(q + (size_t)t * cap, h.data(), (size_t)cap * 16, hipMemcpyHostToDevice);
  const size_t lds = ((size_t)8 << slice_log2) * (MODE == 3 ? THREADS / 64 : 1);
  hipFuncSetAttribute((const void*)apply<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  apply<MODE, THREADS><<<tiles, THREADS, lds>>>(q, n, cap, slice_log2, dt, st);
  hipEventRecord(a);
  apply<MODE, THREADS><<<tiles, THREADS, lds>>>(q, n, cap, slice_log2, dt, st);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long hs[12]; hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
  printf("%-34s tiles=%4d x %5d entries, %6u rec/tile, %4d thr: %7.1f us  (%6.1f GB/s queue)  block0 cycles zero/accum/write = %lld/%lld/%lld\n",
         name, tiles, 1 << slice_log2, n, THREADS, ms * 1e3, (double)tiles * n * 16 / ms / 1e6, hs[0], hs[1], hs[2]);
  hipFree(q); hipFree(dt); hipFree(st);
}
int main() {
  run<0, 1024>("full", 512, 14, 49152);
  run<0, 256>("1K tiles uniform", 640, 10, 65536);
  run<3, 256>("1K tiles uniform, per-wave copies", 640, 10, 65536);
  for (float hf : {0.1f, 0.5f}) for (int hn : {1, 16}) {
    g_hot_frac = hf; g_hot_n = hn;
    printf("-- %.0f%% of the records on %d hot entries\n", hf * 100, hn);
    run<0, 256>("1K tiles", 640, 10, 65536);
    run<3, 256>("1K tiles, per-wave copies", 640, 10, 65536);
    run<0, 1024>("16K tiles", 512, 14, 49152);
  }
  return 0;
}
