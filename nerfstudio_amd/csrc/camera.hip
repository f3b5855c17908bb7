// Camera-pose corrections on the device (gfx950): nsamd_camera_apply, nsamd_camera_backward.
// Reference: cameras/camera_optimizers.py:107-153 (forward, apply_to_raybundle), :179-185 (regulariser),
// cameras/lie_groups.py:25-117 (the exponential maps). Per-camera arithmetic: camera.h.
//
// The reference runs this as ~100 eager torch kernels per step on a [num_cameras, 6] parameter (index, exp map, bmm, and
// their autograd nodes); inside a sub-millisecond training step that is more launches than the whole rest of the iteration.
// Here: one launch forward (a thread per ray evaluates its camera's map — 30 flops, cheaper than staging a table), one launch
// backward (a workgroup per camera sums its rays' dL/dt and dL/dR in double in a fixed order, one thread runs the map's
// closed-form backward; the last workgroup evaluates the regulariser). Bit-reproducible: no float atomics.
#include "camera.h"

namespace nsamd {

constexpr int kCamThreads = 256;

__global__ __launch_bounds__(kCamThreads) void camera_apply_kernel(
    const float* __restrict__ pose, int mode, int num_cameras, const float* __restrict__ raw_o,
    const float* __restrict__ raw_d, const int64_t* __restrict__ cams, int64_t n, float* __restrict__ o,
    float* __restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * kCamThreads + threadIdx.x;
  if (i >= n) return;
  int64_t c = cams[i];
  c = c < 0 ? 0 : (c >= num_cameras ? num_cameras - 1 : c);
  float p[6], R[9], t[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) p[k] = pose[6 * c + k];
  cam_exp_map(mode, p, R, t);
  const float x = raw_d[3 * i], y = raw_d[3 * i + 1], z = raw_d[3 * i + 2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o[3 * i + k] = raw_o[3 * i + k] + t[k];
    d[3 * i + k] = (R[3 * k] * x + R[3 * k + 1] * y) + R[3 * k + 2] * z;
  }
}

// Workgroup c < C: camera c. Workgroup C: the regulariser's value.
__global__ __launch_bounds__(kCamThreads) void camera_backward_kernel(
    const float* __restrict__ pose, int mode, int num_cameras, const float* __restrict__ raw_d,
    const int64_t* __restrict__ cams, int64_t n, nsamd_ray_grads U, float trans_pen, float rot_pen,
    float* __restrict__ dpose, float* __restrict__ reg_out) {
  __shared__ double red[kCamThreads];
  const int tid = threadIdx.x;
  const int c = blockIdx.x;
  if (c == num_cameras) {
    // mean_c |v_c| * trans_pen + mean_c |w_c| * rot_pen
    double sv = 0.0, sw = 0.0;
    for (int k = tid; k < num_cameras; k += kCamThreads) {
      const float* p = pose + 6 * k;
      sv += sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
      sw += sqrt((double)p[3] * p[3] + (double)p[4] * p[4] + (double)p[5] * p[5]);
    }
    double tot[2] = {sv, sw};
    for (int q = 0; q < 2; ++q) {
      red[tid] = tot[q];
      __syncthreads();
      for (int m = kCamThreads / 2; m > 0; m >>= 1) {
        if (tid < m) red[tid] += red[tid + m];
        __syncthreads();
      }
      tot[q] = red[0];
      __syncthreads();
    }
    if (tid == 0 && reg_out != nullptr)
      *reg_out = (float)(tot[0] / num_cameras) * trans_pen + (float)(tot[1] / num_cameras) * rot_pen;
    return;
  }
  // sums over the camera's rays, thread `tid` takes rays tid, tid + 256, ... in order
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.0;
  for (int64_t i = tid; i < n; i += kCamThreads) {
    int64_t ci = cams[i];
    ci = ci < 0 ? 0 : (ci >= num_cameras ? num_cameras - 1 : ci);  // clamped as in the forward (camera_apply_kernel)
    if (ci != c) continue;
    float go[3], gd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float so = 0.0f, sd = 0.0f;  // the levels' shares in their order (the proposal levels first, the main level last)
      for (int l = 0; l < U.count; ++l) {
        so += U.d_origins[l][3 * i + k];
        sd += U.d_directions[l][3 * i + k];
      }
      go[k] = so, gd[k] = sd;
    }
    const float x = raw_d[3 * i], y = raw_d[3 * i + 1], z = raw_d[3 * i + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      acc[3 * k + 0] += (double)gd[k] * x;  // dL/dR[k][j] = sum_rays dL/dd'[k] * d[j]
      acc[3 * k + 1] += (double)gd[k] * y;
      acc[3 * k + 2] += (double)gd[k] * z;
      acc[9 + k] += (double)go[k];
    }
  }
  for (int q = 0; q < 12; ++q) {  // fixed-order tree per component
    red[tid] = acc[q];
    __syncthreads();
    for (int m = kCamThreads / 2; m > 0; m >>= 1) {
      if (tid < m) red[tid] += red[tid + m];
      __syncthreads();
    }
    acc[q] = red[0];
    __syncthreads();
  }
  if (tid == 0) {
    float p[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = pose[6 * c + k];
    double dp[6], dr[6];
    cam_exp_map_bwd(mode, p, acc, acc + 9, dp);
    cam_reg_bwd(p, (double)trans_pen / num_cameras, (double)rot_pen / num_cameras, dr);
#pragma unroll
    for (int k = 0; k < 6; ++k) dpose[6 * c + k] += (float)dp[k] + (float)dr[k];
  }
}

}  // namespace nsamd

using namespace nsamd;

extern "C" int nsamd_camera_apply(const float* pose, int32_t mode, int32_t num_cameras, const float* raw_origins,
                                  const float* raw_directions, const int64_t* camera_indices, int64_t n, float* origins,
                                  float* directions, nsamd_stream_t stream) {
  NSAMD_REQUIRE(n >= 0 && num_cameras > 0 && (mode == kCamSO3xR3 || mode == kCamSE3));
  if (n == 0) return NSAMD_OK;
  NSAMD_REQUIRE(pose && raw_origins && raw_directions && camera_indices && origins && directions);
  camera_apply_kernel<<<(unsigned)((n + kCamThreads - 1) / kCamThreads), kCamThreads, 0, (hipStream_t)stream>>>(
      pose, mode, num_cameras, raw_origins, raw_directions, camera_indices, n, origins, directions);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}

extern "C" int nsamd_camera_backward(const float* pose, int32_t mode, int32_t num_cameras, const float* raw_directions,
                                     const int64_t* camera_indices, int64_t n, nsamd_ray_grads upstream,
                                     float trans_l2_penalty, float rot_l2_penalty, float* dpose, float* regulariser,
                                     nsamd_stream_t stream) {
  NSAMD_REQUIRE(n >= 0 && num_cameras > 0 && (mode == kCamSO3xR3 || mode == kCamSE3));
  NSAMD_REQUIRE(pose && dpose && upstream.count >= 0 && upstream.count <= 4);
  if (n > 0) NSAMD_REQUIRE(raw_directions && camera_indices);
  for (int l = 0; l < upstream.count; ++l) NSAMD_REQUIRE(upstream.d_origins[l] && upstream.d_directions[l]);
  camera_backward_kernel<<<(unsigned)num_cameras + 1u, kCamThreads, 0, (hipStream_t)stream>>>(
      pose, mode, num_cameras, raw_directions, camera_indices, n, upstream, trans_l2_penalty, rot_l2_penalty, dpose,
      regulariser);
  NSAMD_CHECK_LAUNCH();
  return NSAMD_OK;
}
