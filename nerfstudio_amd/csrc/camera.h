// Pose corrections of the training cameras: the exponential maps and their backward, per camera (gfx950; the host compiler
// sees them only through tests/hostcheck). Reference: nerfstudio/cameras/lie_groups.py:25-60 (SO3xR3), :63-117 (SE3),
// cameras/camera_optimizers.py:107-153 (forward / apply_to_raybundle), :179-185 (the L2 regulariser).
//
// Forward in fp32 with the reference's operations in the reference's order (branch points 1e-4 on |w|^2 and 1e-2 on |w|, the
// Taylor forms below them). Backward: closed-form derivatives of exactly those expressions — what autograd differentiates —
// evaluated in double on per-camera sums that were accumulated in double in a fixed order.
#pragma once

#include "common.h"

namespace nsamd {

constexpr int kCamSO3xR3 = 1;
constexpr int kCamSE3 = 2;

// p = (translation v, rotation vector w) -> R (row-major 3x3), t
NSAMD_HD void cam_exp_map(int mode, const float* p, float* R, float* t) {
  const float v0 = p[0], v1 = p[1], v2 = p[2], w0 = p[3], w1 = p[4], w2 = p[5];
  if (mode == kCamSO3xR3) {
    const float nrm = w0 * w0 + w1 * w1 + w2 * w2;
    const float th = sqrtf(fmaxf(nrm, 1e-4f));
    const float inv = 1.0f / th;
    const float a = inv * sinf(th);
    const float b = inv * inv * (1.0f - cosf(th));
    const float K[9] = {0.0f, -w2, w1, w2, 0.0f, -w0, -w1, w0, 0.0f};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const float k2 = K[3 * i + 0] * K[0 + j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
        R[3 * i + j] = (a * K[3 * i + j] + b * k2) + (i == j ? 1.0f : 0.0f);
      }
    t[0] = v0, t[1] = v1, t[2] = v2;
    return;
  }
  const float th = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
  const float t2 = th * th, t3 = t2 * th;
  const bool small = th < 1e-2f;
  const float sn = sinf(th);
  const float cs = small ? 8.0f / (4.0f + t2) - 1.0f : cosf(th);
  const float a = small ? 0.5f * cs + 0.5f : sn / th;
  const float b = small ? 0.5f * a : (1.0f - cs) / t2;
  const float w[3] = {w0, w1, w2};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = b * w[i] * w[j] + (i == j ? cs : 0.0f);
  const float s0 = a * w0, s1 = a * w1, s2 = a * w2;
  R[1] -= s2, R[3] += s2, R[2] += s1, R[6] -= s1, R[5] -= s0, R[7] += s0;
  const float at = small ? 1.0f - t2 / 6.0f : a;
  const float bt = small ? 0.5f - t2 / 24.0f : b;
  const float ct = small ? 1.0f / 6.0f - t2 / 120.0f : (th - sn) / t3;
  const float c0 = w1 * v2 - w2 * v1, c1 = w2 * v0 - w0 * v2, c2 = w0 * v1 - w1 * v0;  // w x v
  const float wv = w0 * v0 + w1 * v1 + w2 * v2;
  t[0] = (at * v0 + bt * c0) + ct * (w0 * wv);
  t[1] = (at * v1 + bt * c1) + ct * (w1 * wv);
  t[2] = (at * v2 + bt * c2) + ct * (w2 * wv);
}

// G = dL/dR (row-major), g = dL/dt of ONE camera -> dp[6] = dL/d(v, w)
NSAMD_HD void cam_exp_map_bwd(int mode, const float* p, const double* G, const double* g, double* dp) {
  const double v[3] = {p[0], p[1], p[2]}, w[3] = {p[3], p[4], p[5]};
  // vee of the antisymmetric part: <G, hat(u)> = u . skew(G)
  const double skew[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};
  if (mode == kCamSO3xR3) {
    const double nrm = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const bool clamped = !(nrm >= 1e-4f);  // torch.clamp passes the gradient where the input is >= min
    const double th = sqrt(clamped ? (double)1e-4f : nrm);
    const double sn = sin(th), cs = cos(th);
    const double a = sn / th, b = (1.0 - cs) / (th * th);
    const double K[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
    // R = a K + b K^2 + I:  dL/dK = a G + b (G K^T + K^T G)
    double dK[9], dLa = 0.0, dLb = 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double gkT = 0.0, kTg = 0.0, k2 = 0.0;
        for (int k = 0; k < 3; ++k) {
          gkT += G[3 * i + k] * K[3 * j + k];  // (G K^T)[i][j]
          kTg += K[3 * k + i] * G[3 * k + j];  // (K^T G)[i][j]
          k2 += K[3 * i + k] * K[3 * k + j];
        }
        dK[3 * i + j] = a * G[3 * i + j] + b * (gkT + kTg);
        dLa += G[3 * i + j] * K[3 * i + j];
        dLb += G[3 * i + j] * k2;
      }
    double dw[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    if (!clamped) {
      const double da = (cs - a) / th, db = (a - 2.0 * b) / th;
      const double dth = dLa * da + dLb * db;
      for (int i = 0; i < 3; ++i) dw[i] += dth * w[i] / th;
    }
    dp[0] = g[0], dp[1] = g[1], dp[2] = g[2], dp[3] = dw[0], dp[4] = dw[1], dp[5] = dw[2];
    return;
  }
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double t2 = th * th, t3 = t2 * th;
  const bool small = (float)th < 1e-2f;
  const double sn = sin(th);
  double cs, a, b, at, bt, ct, dcs, da, db, dat, dbt, dct;
  if (small) {
    cs = 8.0 / (4.0 + t2) - 1.0, a = 0.5 * cs + 0.5, b = 0.5 * a;
    at = 1.0 - t2 / 6.0, bt = 0.5 - t2 / 24.0, ct = 1.0 / 6.0 - t2 / 120.0;
    dcs = -16.0 * th / ((4.0 + t2) * (4.0 + t2)), da = 0.5 * dcs, db = 0.5 * da;
    dat = -th / 3.0, dbt = -th / 12.0, dct = -th / 60.0;
  } else {
    cs = cos(th), a = sn / th, b = (1.0 - cs) / t2, at = a, bt = b, ct = (th - sn) / t3;
    dcs = -sn, da = (cs - a) / th, db = (sn - 2.0 * b * th) / t2, dat = da, dbt = db;
    dct = ((1.0 - cs) - 3.0 * ct * t2) / t3;
  }
  // R = b w w^T + cs I + hat(a w)
  double wGw = 0.0, Gsw[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      wGw += w[i] * G[3 * i + j] * w[j];
      Gsw[i] += (G[3 * i + j] + G[3 * j + i]) * w[j];
    }
  const double trG = G[0] + G[4] + G[8];
  const double w_skew = w[0] * skew[0] + w[1] * skew[1] + w[2] * skew[2];
  // t = at v + bt (w x v) + ct w (w . v)
  const double wxv[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
  const double gxw[3] = {g[1] * w[2] - g[2] * w[1], g[2] * w[0] - g[0] * w[2], g[0] * w[1] - g[1] * w[0]};
  const double vxg[3] = {v[1] * g[2] - v[2] * g[1], v[2] * g[0] - v[0] * g[2], v[0] * g[1] - v[1] * g[0]};
  const double gv = g[0] * v[0] + g[1] * v[1] + g[2] * v[2];
  const double gw = g[0] * w[0] + g[1] * w[1] + g[2] * w[2];
  const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
  const double g_wxv = g[0] * wxv[0] + g[1] * wxv[1] + g[2] * wxv[2];
  double dth = wGw * db + trG * dcs + w_skew * da + gv * dat + g_wxv * dbt + gw * wv * dct;
  for (int i = 0; i < 3; ++i) {
    dp[i] = at * g[i] + bt * gxw[i] + ct * gw * w[i];
    double dwi = b * Gsw[i] + a * skew[i] + bt * vxg[i] + ct * (g[i] * wv + v[i] * gw);
    if (th > 0.0) dwi += dth * w[i] / th;  // the 2-norm's gradient at the origin is taken as zero (torch masks it)
    dp[3 + i] = dwi;
  }
}

// the regulariser's gradient for one camera: trans_pen * v / |v| / C + rot_pen * w / |w| / C (zero where the norm is zero)
NSAMD_HD void cam_reg_bwd(const float* p, double trans_scale, double rot_scale, double* dp) {
  const double nv = sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
  const double nw = sqrt((double)p[3] * p[3] + (double)p[4] * p[4] + (double)p[5] * p[5]);
  for (int i = 0; i < 3; ++i) {
    dp[i] = nv > 0.0 ? trans_scale * p[i] / nv : 0.0;
    dp[3 + i] = nw > 0.0 ? rot_scale * p[3 + i] / nw : 0.0;
  }
}

}  // namespace nsamd
