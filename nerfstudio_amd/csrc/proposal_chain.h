// Internal interface of the merged proposal-level backward (proposal_chain.hip: nsamd_proposal_levels_bwd): every stage of
// two levels' chains as ONE launch. Each `*_launch_pair` returns NSAMD_ERR_UNSUPPORTED with nothing enqueued when its two
// calls cannot share a launch; the caller then issues them one after the other. Not part of the C ABI.
#pragma once

#include "common.h"

namespace nsamd {

// RaySamples.get_weights backward with the zero-gradient gate (sampler.hip: weights_bwd_kernel)
struct WeightsBwdCall {
  const float* t_bins;
  const float* density;
  const float* dweights;
  int64_t num_rays;
  int32_t S;
  float* ddensity;
  uint32_t* gate;
  uint8_t* ray_mask;
};
int weights_bwd_launch_pair(const WeightsBwdCall& a, const WeightsBwdCall& b, hipStream_t stream);

// density MLP backward + the fixed-order reduce of its weight-gradient partials (density_mlp.hip)
struct DensityBwdCall {
  const float* enc;
  const float* selector;
  const float* pre;
  const float* ddensity;
  int64_t M;
  nsamd_density_mlp mlp;
  float* denc;
  float* dW0;
  float* db0;
  float* dW1;
  float* db1;
  float* workspace;
  int64_t workspace_floats;
  const uint32_t* gate;
  const uint8_t* ray_mask;
  int spr;
};
int density_bwd_launch_pair(const DensityBwdCall& a, const DensityBwdCall& b, hipStream_t stream);

}  // namespace nsamd
