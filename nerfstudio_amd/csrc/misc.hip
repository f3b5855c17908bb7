// Pinhole ray generation, fused Adam and library introspection for gfx950.
#include <string.h>

#include "common.h"

namespace nsamd {

// RayGenerator.forward (/root/reference/nerfstudio/model_components/ray_generators.py:41-56) ->
// Cameras._generate_rays_from_coords, perspective branch (cameras/cameras.py:598-634 coords, :655-656 y flip,
// :781-787 directions, :887-909 rotate, normalise, pixel area). One ray per lane; the per-camera pose and
// intrinsics are gathered through L2 (a few hundred cameras at most).
__device__ __forceinline__ void raygen_pinhole_one(float x, float y, float fxr, float fyr, float cxr, float cyr,
                                                   const float* __restrict__ m, float* __restrict__ origin,
                                                   float* __restrict__ direction, float* __restrict__ pixel_area,
                                                   float* __restrict__ direction_norm) {
  const float eps = 8.881784197001252e-16f;  // camera_utils._EPS = 4 * float64 eps, cast to fp32
  // three coords: centre, +1 in x, +1 in y  (cameras.py:622-634)
  const float px[3] = {(x - cxr) / fxr, (x - cxr + 1.0f) / fxr, (x - cxr) / fxr};
  const float py[3] = {(y - cyr) / fyr, (y - cyr) / fyr, (y - cyr + 1.0f) / fyr};
  float d[3][3];
  float n0 = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lx = px[k], ly = -py[k], lz = -1.0f;  // OpenCV -> OpenGL (cameras.py:655-656), z = -1 (:787)
    float v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = (lx * m[4 * r + 0] + ly * m[4 * r + 1]) + lz * m[4 * r + 2];
    const float nrm = fmaxf(sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]), eps);
    if (k == 0) n0 = nrm;
#pragma unroll
    for (int r = 0; r < 3; ++r) d[k][r] = v[r] / nrm;
  }
  float dx = 0.0f, dy = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float a = d[0][r] - d[1][r], b = d[0][r] - d[2][r];
    dx += a * a;
    dy += b * b;
  }
  dx = sqrtf(dx);
  dy = sqrtf(dy);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    origin[r] = m[4 * r + 3];
    direction[r] = d[0][r];
  }
  if (pixel_area) *pixel_area = dx * dy;
  if (direction_norm) *direction_norm = n0;
}

__global__ void raygen_pinhole_kernel(const int64_t* __restrict__ ray_indices, const float* __restrict__ c2w,
                                      const float* __restrict__ fx, const float* __restrict__ fy,
                                      const float* __restrict__ cx, const float* __restrict__ cy, int64_t num_rays,
                                      float* __restrict__ origins, float* __restrict__ directions,
                                      float* __restrict__ pixel_area, float* __restrict__ directions_norm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_rays) return;
  const int64_t cam = ray_indices[3 * i + 0];
  const float y = (float)ray_indices[3 * i + 1] + 0.5f;  // image_coords = meshgrid + 0.5 (cameras.py:312-313)
  const float x = (float)ray_indices[3 * i + 2] + 0.5f;
  raygen_pinhole_one(x, y, fx[cam], fy[cam], cx[cam], cy[cam], c2w + cam * 12, origins + 3 * i, directions + 3 * i, pixel_area + i,
                     directions_norm ? directions_norm + i : nullptr);
}

// The rays of ONE camera's image in row-major pixel order, generated where they are consumed: ray i of the launch is pixel
// first_pixel + i of the implicit (row, col) grid — `Cameras.generate_rays(camera_indices=0, keep_shape=True)` of
// Model.get_outputs_for_camera (models/base_model.py:166-175) restricted to the chunk the render loop is about to trace, so no
// [H, W, 3] bundle (nor an index list) exists in HBM. Rays past `num_rays` (the padding of a last, shorter chunk) repeat the
// last pixel. Same arithmetic as raygen_pinhole_kernel: the same bits as indexing the full bundle.
__global__ void raygen_pinhole_grid_kernel(const float* __restrict__ c2w, float fx, float fy, float cx, float cy, int32_t width,
                                           int64_t first_pixel, int64_t num_rays, int64_t padded, float* __restrict__ origins,
                                           float* __restrict__ directions, float* __restrict__ pixel_area) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  const int64_t p = first_pixel + (i < num_rays ? i : num_rays - 1);
  const int64_t row = p / width, col = p - row * width;
  raygen_pinhole_one((float)col + 0.5f, (float)row + 0.5f, fx, fy, cx, cy, c2w, origins + 3 * i, directions + 3 * i,
                     pixel_area ? pixel_area + i : nullptr, nullptr);
}

// The step's ray batch out of a pool of pre-generated batches resident in HBM: what VanillaDataManager.next_train
// (/root/reference/nerfstudio/data/datamanagers/base_datamanager.py:506-515) hands the model each iteration. The slot
// comes from DEVICE memory (a float, as the other per-step scalars) so that a captured hipGraph replays with a new
// batch every step. One launch copies origins, directions, camera indices and target colours.
__global__ void select_batch_kernel(const float* __restrict__ slot_dev, int32_t slots, int64_t n,
                                    const float* __restrict__ origins_pool, const float* __restrict__ directions_pool,
                                    const int64_t* __restrict__ cameras_pool, const float* __restrict__ target_pool,
                                    float* __restrict__ origins, float* __restrict__ directions,
                                    int64_t* __restrict__ cameras, float* __restrict__ target) {
  int32_t slot = (int32_t)slot_dev[0];
  slot = slot < 0 ? 0 : (slot >= slots ? slots - 1 : slot);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * n) {
    const int64_t src = (int64_t)slot * 3 * n + i;
    origins[i] = origins_pool[src];
    directions[i] = directions_pool[src];
    target[i] = target_pool[src];
  }
  if (i < n) cameras[i] = cameras_pool[(int64_t)slot * n + i];
}

// Compact exchange of a hash-table prefix (arena.ParamArena.register_compact): the rows of the coarse levels that a
// position can ever reach (functional.HashGridSpec.reachable_prefix) are packed into one dense buffer before the
// all-reduce and unpacked after it. rows [*, feat] fp32, index [n] int64 (sorted), feat == 2 for the hash tables.
__global__ void rows_gather_kernel(const float* __restrict__ rows, const int64_t* __restrict__ index, int64_t n, int feat,
                                   float* __restrict__ packed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * feat) return;
  const int64_t r = i / feat;
  packed[i] = rows[index[r] * feat + (i - r * feat)];
}

__global__ void rows_scatter_kernel(float* __restrict__ rows, const int64_t* __restrict__ index, int64_t n, int feat,
                                    const float* __restrict__ packed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * feat) return;
  const int64_t r = i / feat;
  rows[index[r] * feat + (i - r * feat)] = packed[i];
}

// Head of a captured training iteration: everything that changes from step to step, produced ON THE DEVICE so that a replay
// needs no host-issued operation in front of it (an eager 32-byte upload and torch's own generator-offset fill in front of every
// hipGraphLaunch cost ~8 us of idle stream each — profiles/r05_s9_seam_trace_gaps.txt):
//   * the step's scalars (Adam step sizes of every optimiser group, the anneal exponent, the batch slot): row counter[0] % rows
//     of a table of rows in exact host arithmetic (nothing is re-derived here) — either device memory the host uploaded ahead
//     for the coming iterations, or a ring in pinned host memory the host fills one row per launch, read over the bus;
//   * the step's uniform draws (the sampler's jitter per level and ray, the loss's random background): Philox-4x32-10 keyed by
//     (seed, counter[1]) — a counter-based generator needs no state beyond the step number, so eager launches and replays of
//     the same step draw the same numbers.
// One workgroup: the counters are read by every thread before thread 0 advances them.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
  c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
  k[0] += 0x9E3779B9u, k[1] += 0xBB67AE85u;
}

__global__ __launch_bounds__(1024) void step_prologue_kernel(int64_t* __restrict__ counter, const float* __restrict__ table,
                                                             int rows, float* __restrict__ hyper, float* __restrict__ out0,
                                                             int64_t n0, float* __restrict__ out1, int64_t n1, uint64_t seed) {
  const int64_t row = counter[0], draw = counter[1];
  // (system-scope load: the rows may lie in pinned host memory the host wrote just before the launch — never a cached copy)
  if (table != nullptr && rows > 0 && threadIdx.x < 8)
    hyper[threadIdx.x] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(table) + (row % rows) * 8 + threadIdx.x,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  const int64_t total = n0 + n1;
  for (int64_t q = threadIdx.x; 4 * q < total; q += 1024) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)draw, (uint32_t)(draw >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = 4 * q + j;
      const float u = (float)(c[j] >> 8) * 5.9604644775390625e-8f;  // 24 random bits: uniform on [0, 1)
      if (i < n0) out0[i] = u;
      else if (i < total) out1[i - n0] = u;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    counter[0] = row + 1;
    counter[1] = draw + 1;
  }
}

// torch.optim.Adam (no amsgrad / weight decay / maximize): one pass over the flat arena, 16 B per lane — with the operation
// order of torch/optim/adam.py _single_tensor_adam on fp32 tensors, so that the update is the SAME BITS as torch's:
//   exp_avg.lerp_(grad, 1 - beta1)                          fma(w1, g - m, m),   w1 = float(1 - beta1) (double subtraction)
//   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)  fma(w2 g, g, v beta2)
//   denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   param.addcdiv_(exp_avg, denom, value=-step_size)        p + ((-step_size) m) / denom
// step_size = lr / (1 - b1^t) and bc2_sqrt = sqrt(1 - b2^t) are evaluated in double by the host and rounded once, as the
// Python scalars are when ATen takes them; correctly rounded sqrt and division (hipcc default), no FMA contraction.
// The moments (2 x 67 MB for the main table) are touched exactly once per step: streamed past the caches with
// non-temporal accesses, so that the parameters and the gradient — which the next launches read again — keep their place
// (NSAMD_ADAM_NT=0 at build time: plain accesses, for A/B).
#ifndef NSAMD_ADAM_NT
#define NSAMD_ADAM_NT 1
#endif
typedef float adam_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load4(const float* base, int64_t i) {
#if NSAMD_ADAM_NT
  const adam_v4 v = __builtin_nontemporal_load(reinterpret_cast<const adam_v4*>(base) + i);
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return reinterpret_cast<const float4*>(base)[i];
#endif
}
__device__ __forceinline__ void stream_store4(float* base, int64_t i, const float4& x) {
#if NSAMD_ADAM_NT
  __builtin_nontemporal_store(adam_v4{x.x, x.y, x.z, x.w}, reinterpret_cast<adam_v4*>(base) + i);
#else
  reinterpret_cast<float4*>(base)[i] = x;
#endif
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float beta2, float w1, float w2, float eps,
                            float step_size_host, float bc2_sqrt_host, const float* __restrict__ hyper_dev,
                            float grad_scale) {
  const float neg_step = -(hyper_dev ? hyper_dev[0] : step_size_host);
  const float bc2_sqrt = hyper_dev ? hyper_dev[1] : bc2_sqrt_host;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = stream_load4(m, i);
    float4 vv = stream_load4(v, i);
    // Rows that never received a gradient (g = m = v = 0) get a zero update: leave them alone. The torch path hashes
    // the coarse levels too, which can only ever reach (res + 1)^3 of their 2^19 slots (332 k of 2.6 M
    // rows on levels 0-4) — whole cache lines of the arena are never touched, and this skips their p read and the
    // three writes.
    if (gg.x == 0.0f && gg.y == 0.0f && gg.z == 0.0f && gg.w == 0.0f && mm.x == 0.0f && mm.y == 0.0f && mm.z == 0.0f &&
        mm.w == 0.0f && vv.x == 0.0f && vv.y == 0.0f && vv.z == 0.0f && vv.w == 0.0f)
      continue;
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float* pa = reinterpret_cast<float*>(&pp);
    const float* ga = reinterpret_cast<const float*>(&gg);
    float* ma = reinterpret_cast<float*>(&mm);
    float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = grad_scale == 1.0f ? ga[k] : ga[k] * grad_scale;
      ma[k] = fmaf(w1, gr - ma[k], ma[k]);  // ATen's lerp kernel is a fused multiply-add (cpu/LerpKernel.cpp lerp_vec)
      va[k] = fmaf(w2 * gr, gr, va[k] * beta2);  // addcmul's (alpha t1) t2 + self is contracted in ATen's vector kernel
      const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
      pa[k] = pa[k] + (neg_step * ma[k]) / denom;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    stream_store4(m, i, mm);
    stream_store4(v, i, vv);
  }
  // tail (n not a multiple of 4)
  const int64_t tail0 = n4 << 2;
  const int64_t t = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float gr = grad_scale == 1.0f ? g[t] : g[t] * grad_scale;
    const float m1 = fmaf(w1, gr - m[t], m[t]);
    const float v1 = fmaf(w2 * gr, gr, v[t] * beta2);
    m[t] = m1;
    v[t] = v1;
    p[t] = p[t] + (neg_step * m1) / (sqrtf(v1) / bc2_sqrt + eps);
  }
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_raygen_pinhole(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy,
                                    const float* cx, const float* cy, int64_t num_rays, int32_t num_cameras,
                                    float* origins, float* directions, float* pixel_area, float* directions_norm,
                                    nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && num_cameras > 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(ray_indices && c2w && fx && fy && cx && cy && origins && directions && pixel_area);
  raygen_pinhole_kernel<<<(unsigned)((num_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      ray_indices, c2w, fx, fy, cx, cy, num_rays, origins, directions, pixel_area, directions_norm);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_raygen_pinhole_grid(const float* c2w, float fx, float fy, float cx, float cy, int32_t width,
                                         int64_t first_pixel, int64_t num_rays, int64_t padded_rays, float* origins,
                                         float* directions, float* pixel_area, nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && padded_rays >= num_rays && width > 0 && first_pixel >= 0);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(c2w && origins && directions && fx != 0.0f && fy != 0.0f);
  raygen_pinhole_grid_kernel<<<(unsigned)((padded_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      c2w, fx, fy, cx, cy, width, first_pixel, num_rays, padded_rays, origins, directions, pixel_area);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_rows_gather(const float* rows, const int64_t* index, int64_t n, int32_t feat, float* packed,
                                 nsamd_stream_t stream) {
  NSAMD_REQUIRE(n >= 0 && feat >= 1);
  if (n == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rows && index && packed);
  const int64_t nb = (n * feat + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  rows_gather_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(rows, index, n, feat, packed);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_rows_scatter(float* rows, const int64_t* index, int64_t n, int32_t feat, const float* packed,
                                  nsamd_stream_t stream) {
  NSAMD_REQUIRE(n >= 0 && feat >= 1);
  if (n == 0) return NSAMD_OK;
  NSAMD_REQUIRE(rows && index && packed);
  const int64_t nb = (n * feat + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  rows_scatter_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(rows, index, n, feat, packed);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_select_batch(const float* slot_dev, int32_t slots, int64_t num_rays, const float* origins_pool,
                                  const float* directions_pool, const int64_t* cameras_pool, const float* target_pool,
                                  float* origins, float* directions, int64_t* cameras, float* target,
                                  nsamd_stream_t stream) {
  NSAMD_REQUIRE(num_rays >= 0 && slots >= 1);
  if (num_rays == 0) return NSAMD_OK;
  NSAMD_REQUIRE(slot_dev && origins_pool && directions_pool && cameras_pool && target_pool && origins && directions &&
                cameras && target);
  const int64_t nb = (3 * num_rays + 255) / 256;
  if (nb > 0x7fffffffLL) return NSAMD_ERR_UNSUPPORTED;
  select_batch_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(slot_dev, slots, num_rays, origins_pool,
                                                                     directions_pool, cameras_pool, target_pool,
                                                                     origins, directions, cameras, target);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_step_prologue(int64_t* counter, const float* table, int32_t rows, float* hyper, float* uniform0,
                                   int64_t n0, float* uniform1, int64_t n1, uint64_t seed, nsamd_stream_t stream) {
  NSAMD_REQUIRE(counter != nullptr && n0 >= 0 && n1 >= 0 && rows >= 0);
  NSAMD_REQUIRE(table == nullptr || (rows > 0 && hyper != nullptr));
  NSAMD_REQUIRE((n0 == 0 || uniform0 != nullptr) && (n1 == 0 || uniform1 != nullptr));
  step_prologue_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(counter, table, rows, hyper, uniform0, n0, uniform1, n1, seed);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                               double lr, double beta1, double beta2, double eps, int32_t step, float grad_scale,
                               const float* hyper_dev, nsamd_stream_t stream) {
  NSAMD_REQUIRE(n >= 0 && step >= 1);
  if (n == 0) return NSAMD_OK;
  NSAMD_REQUIRE(params && grads && exp_avg && exp_avg_sq);
  NSAMD_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int64_t n4 = (n + 3) / 4;
  // Workgroups of the grid-stride pass: 4 per CU (NSAMD_ADAM_BLOCKS_PER_CU). Alone the 470 MB main group takes 61.8 / 62.0 / 61.3 /
  // 67.3 us with 8 / 4 / 3 / 2 per CU — HBM is saturated from 3 on — but the deferred main-field pass runs BESIDE the next
  // proposal forward, and with 8 per CU it starves it (the launch that writes the initial bins reads 45 us there, 14 us alone):
  // driver window 0.7231 (8) / 0.7150 (4) / 0.7142 ms (2), three alternating repeats on one box, profiles/r05_s21_*, r05_s22_*.
  // Round 6 (profiles/r06_s18_*, three alternating repeats, window / 300-step long run): 4: 0.6532 / 0.7088, 3: 0.6486 / 0.7030,
  // 2: 0.6533 / 0.7042, 1: 0.6586 / 0.7051 ms -> 3.
  static const int per_cu = [] { const char* e = getenv("NSAMD_ADAM_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 3; return v > 0 ? v : 3; }();
  const unsigned blocks = (unsigned)min((int64_t)256 * per_cu, (n4 + 255) / 256);
  adam_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, (float)beta2,
                                                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, step_size,
                                                       bc2_sqrt, hyper_dev, grad_scale);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" const char* nsamd_version(void) { return "nsamd 0.1.0 (gfx950)"; }

extern "C" const char* nsamd_status_string(int status) {
  switch (status) {
    case NSAMD_OK: return "ok";
    case NSAMD_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or inconsistent shapes)";
    case NSAMD_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
    case NSAMD_ERR_LAUNCH: return "HIP kernel launch failed";
    case NSAMD_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown nsamd status";
  }
}

extern "C" int nsamd_device_info(int32_t* num_cus, int32_t* wavefront_size, int32_t* lds_bytes_per_cu,
                                 char* arch_name, int32_t arch_name_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return NSAMD_ERR_NO_DEVICE;
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (wavefront_size) *wavefront_size = prop.warpSize;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
    arch_name[arch_name_len - 1] = '\0';
  }
  return NSAMD_OK;
}
