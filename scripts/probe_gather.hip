// GPU diagnostic (not part of the product): what a divergent 16-B gather costs on gfx950 as a function of how the lanes of a
// wave share 128-B lines — the question behind the hash forward's fine levels (91 % TA busy, ~1 line per 2 clocks per CU).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_gather.hip -o scripts/probe_gather.bin && scripts/probe_gather.bin
// Table: 4 MB (one level slice of the nerfacto main table; L2-resident). Every thread does ITER dependent-free gathers of one
// float4 (or float2) at pseudo-random entries; GROUP adjacent lanes draw the SAME 128-B line (different 16-B chunks of it).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int GROUP, int WIDTH>  // GROUP lanes share a line; WIDTH 16 or 8 bytes per lane
__global__ __launch_bounds__(256) void gather(const float* __restrict__ table, uint32_t line_mask, int iters, float* out) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t grp = tid / GROUP, sub = tid % GROUP;
  uint32_t s = grp * 2654435761u + 12345u;
  float acc = 0.f;
#pragma unroll 8
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const uint32_t line = (s >> 7) & line_mask;                         // 128-B line of the table
    const uint32_t chunk = (sub + (s >> 3)) & 7u;                       // 16-B chunk inside it (distinct per lane of a group up to 8)
    const char* p = reinterpret_cast<const char*>(table) + (size_t)line * 128u + chunk * 16u;
    if (WIDTH == 16) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      acc += v.x + v.y + v.z + v.w;
    } else {
      const float2 v = *reinterpret_cast<const float2*>(p);
      acc += v.x + v.y;
    }
  }
  if (acc == 123456.789f) out[tid] = acc;  // never true: keeps the loads
}

template <int GROUP, int WIDTH>
static void run(const float* table, float* out, uint32_t line_mask, const char* what) {
  const int blocks = 256 * 16, iters = 64;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  gather<GROUP, WIDTH><<<blocks, 256>>>(table, line_mask, iters, out);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) gather<GROUP, WIDTH><<<blocks, 256>>>(table, line_mask, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double lane_ops = 5.0 * blocks * 256.0 * iters;
  const double us = ms * 1e3 / 5.0;
  // per CU and clock (2.1 GHz under load, 256 CUs)
  printf("%-44s %8.1f us  %6.2f G lane-gathers/s  %5.3f lane-gathers/clk/CU  %5.3f lines/clk/CU\n", what, us, lane_ops / (ms * 1e-3) / 1e9,
         lane_ops / (ms * 1e-3) / 256.0 / 2.1e9, lane_ops / GROUP / (ms * 1e-3) / 256.0 / 2.1e9);
}

int main() {
  const size_t bytes = 4u << 20;
  float *table, *out;
  hipMalloc(&table, bytes);
  hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  hipMemset(table, 0, bytes);
  const uint32_t line_mask = (uint32_t)(bytes / 128) - 1u;
  run<1, 16>(table, out, line_mask, "16 B, every lane its own line");
  run<2, 16>(table, out, line_mask, "16 B, lane pairs share a line");
  run<4, 16>(table, out, line_mask, "16 B, quads share a line");
  run<8, 16>(table, out, line_mask, "16 B, 8 lanes share a line (all of it)");
  run<1, 8>(table, out, line_mask, " 8 B, every lane its own line");
  run<2, 8>(table, out, line_mask, " 8 B, lane pairs share a line");
  run<4, 8>(table, out, line_mask, " 8 B, quads share a line");
  // a larger table (64 MB: the whole main table, mostly L2 misses -> Infinity Cache / HBM)
  float* big;
  hipMalloc(&big, 64u << 20);
  hipMemset(big, 0, 64u << 20);
  const uint32_t big_mask = (uint32_t)((64u << 20) / 128) - 1u;
  run<1, 16>(big, out, big_mask, "16 B, own line, 64 MB table");
  run<2, 16>(big, out, big_mask, "16 B, pairs share a line, 64 MB table");
  return 0;
}
