/*
 * nsamd.h — C ABI of libnsamd.so: the MI355X (gfx950) volumetric-rendering core behind nerfstudio's
 * Field / Encoding / Sampler / Renderer plugin API (nerfacto hot path).
 *
 * The boundary: plain pointers, sizes, small POD structs passed by value and a hipStream_t. No torch types.
 * Every pointer is a DEVICE pointer unless the name ends in `_host`. All tensors are fp32 unless noted, dense,
 * row-major. Every entry point returns 0 (NSAMD_OK) or a negative nsamd_status; nothing is launched on error.
 * Kernels are enqueued on `stream` and never synchronise the device.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference/nerfstudio/):
 *   the reference's only FFI seam for this path is `implementation="tcnn"` —
 *     tcnn.Encoding(HashGrid)            field_components/encodings.py:353-365, :460-463
 *     tcnn.Encoding(SphericalHarmonics)  field_components/encodings.py:772-786, :796-799
 *     tcnn.Network / NetworkWithInputEncoding   field_components/mlp.py:103-114, :252-269, :181-184, :294-295
 *   and nerfacc for packed compositing (model_components/renderers.py:97-102). Everything else on the path is
 *   eager torch; those stages are listed per function below with the torch-path lines they restate.
 *
 * Numerics contract (BASELINE.json north_star): results follow the reference's TORCH path —
 * bit-exact integer sample indices and bins for identical inputs (left-to-right fp32 sums, IEEE div, no FMA
 * contraction in the sampler kernels), fp32 everywhere, RGB within 1e-4 L-inf of the torch field.
 */
#ifndef NSAMD_H
#define NSAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without pulling hip headers into C callers (ctypes passes the raw pointer value). */
typedef void* nsamd_stream_t;

typedef enum nsamd_status {
  NSAMD_OK = 0,
  NSAMD_ERR_INVALID_ARG = -1,   /* null pointer, negative size, inconsistent shapes      */
  NSAMD_ERR_UNSUPPORTED = -2,   /* configuration outside what the kernels are built for  */
  NSAMD_ERR_LAUNCH = -3,        /* hipGetLastError() != hipSuccess after the launch      */
  NSAMD_ERR_NO_DEVICE = -4      /* no gfx950 device / HIP runtime error at query time    */
} nsamd_status;

#define NSAMD_MAX_LEVELS 32

/* Multiresolution hash grid (HashEncoding ctor, field_components/encodings.py:321-370).
 * features_per_level is fixed at 2 (the only value any nerfacto config uses); table is [L * 2^log2_T, 2].
 * scalings[l] = floor(min_res * growth^l) evaluated in fp32 by the host exactly as the reference does. */
typedef struct nsamd_grid {
  int32_t num_levels;
  int32_t log2_table_size;
  float scalings[NSAMD_MAX_LEVELS];
} nsamd_grid;

/* Where the M sample points come from. Either
 *   positions != NULL : explicit [M,3] positions (Field.density_fn, fields/base_field.py:48-68), or
 *   positions == NULL : M = num_rays * S points o + d * (t[s] + t[s+1]) / 2   (Frustums.get_positions,
 *                       cameras/rays.py:50-59) — positions are never materialised in HBM.               */
typedef struct nsamd_points {
  const float* positions;  /* [M,3] or NULL                    */
  const float* origins;    /* [num_rays,3]      (ray mode)     */
  const float* directions; /* [num_rays,3]      (ray mode)     */
  const float* t_bins;     /* [num_rays,S+1]    (ray mode)     */
  int32_t samples_per_ray; /* S                 (ray mode)     */
} nsamd_points;

/* Position normalisation ahead of the hash grid (fields/density_fields.py:95-103, nerfacto_field.py:205-214). */
typedef enum nsamd_transform {
  NSAMD_XFORM_NONE = 0,       /* x already in [0,1]^3; selector = 1                                            */
  NSAMD_XFORM_CONTRACT = 1,   /* L-inf SceneContraction (spatial_distortions.py:66-69), (x+2)/4, selector mask */
  NSAMD_XFORM_AABB = 2        /* (x - aabb_min) / (aabb_max - aabb_min)  (data/scene_box.py:62-71), selector   */
} nsamd_transform;

typedef struct nsamd_aabb { float lo[3]; float hi[3]; } nsamd_aabb;

/* ------------------------------------------------------------------------------------------------------------
 * Hash encoding.  Replaces tcnn.Encoding(HashGrid) with TORCH-path semantics (HashEncoding.pytorch_fwd,
 * encodings.py:417-458; hash_fn :398-415): ceil/floor corners, every level hashed, blend order x,y,z.
 * enc element (point p, feature k = 2*level + f) is written at enc[p*stride_p + k*stride_k]; the fused fields use
 * the feature-major layout (stride_p = 1, stride_k = M) so that both producer and consumer are coalesced.
 * selector (nullable) receives 1.0f / 0.0f per point (all coords strictly inside (0,1) after the transform).
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_hashgrid_encode_fwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                              nsamd_grid grid, float* enc, int64_t stride_p, int64_t stride_k, float* selector,
                              nsamd_stream_t stream);

/* Backward: dtable[L*T,2] += scatter of denc (caller zero-fills or accumulates); dpositions (nullable) [M,3]
 * receives dL/d(raw position) through offset = scaled - floor(scaled), the selector, the affine map and the
 * contraction Jacobian exactly as autograd differentiates the reference (SURVEY.md §8a gradient-flow facts).
 * `workspace` (nullable, `workspace_floats` 4-byte words of device scratch, base 16-B aligned) enables the binned
 * two-pass scatter (csrc/scatter.hip): no float atomics, sums accumulated in 64-bit fixed point, so the result is
 * BIT-REPRODUCIBLE from run to run (the reference's CPU index_put is sequential; its CUDA one is not reproducible).
 * The workspace must hold nsamd_hashgrid_encode_bwd_workspace(grid, M, write_only) words and its first
 * nsamd_hashgrid_encode_bwd_workspace_state(grid, M) words must be ZERO before the first call (every call leaves them
 * zero again; the rest needs no initialisation). One workspace serves one call at a time (no concurrent streams).
 * Smaller or NULL selects a scratch-free path (float atomics / racing LDS sums: same values up to summation order). */
int nsamd_hashgrid_encode_bwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                              nsamd_grid grid, const float* denc, int64_t stride_p, int64_t stride_k,
                              float* dtable, float* dpositions, float* workspace, int64_t workspace_floats,
                              nsamd_stream_t stream);

/* Same scatter, but dtable is WRITE-ONLY: every entry is set (zero where nothing lands), so the caller neither
 * zero-fills the gradient before the call nor pays the read of the accumulate — with Adam consuming the gradient right
 * after, that removes 2 x 67 MB of HBM traffic per step for the nerfacto main table. Needs the workspace of
 * nsamd_hashgrid_encode_bwd_workspace(grid, M, 1) (its spill list is sized for the worst case, so the result never
 * depends on how the updates spread over the table); otherwise it zero-fills and accumulates. */
int nsamd_hashgrid_encode_bwd_set(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                                  nsamd_grid grid, const float* denc, int64_t stride_p, int64_t stride_k, float* dtable,
                                  float* dpositions, float* workspace, int64_t workspace_floats, nsamd_stream_t stream);

/* dL/d(origins, directions) of the RAYS behind the sample points (ray mode only: pts.positions == NULL): the position
 * gradient of nsamd_hashgrid_encode_bwd's `dpositions`, reduced per ray on the device — d_origins[r] = sum_s dL/dp_s,
 * d_directions[r] = sum_s dL/dp_s * (t_s + t_{s+1}) / 2 (positions = o + d (start + end) / 2, cameras/rays.py:50-59).
 * This is the gradient the camera optimiser consumes (cameras/camera_optimizers.py:148-153: origins and directions are
 * functions of the per-camera pose correction; nerfacto's default mode is SO3xR3, method_configs.py:102). [N,3] each,
 * written (accumulate = 0) or added to (accumulate = 1, e.g. over the proposal levels and the main field). Fixed
 * summation order: bit-reproducible. */
int nsamd_hashgrid_encode_bwd_rays(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                                   nsamd_grid grid, const float* denc, int64_t stride_p, int64_t stride_k,
                                   float* d_origins, float* d_directions, int accumulate, nsamd_stream_t stream);

/* ---- zero-gradient gating of a proposal level's backward chain ---------------------------------------------------
 * The proposal networks receive gradient only through interlevel_loss (model_components/losses.py:113-131; the main
 * weights are detached, :119-120), and that loss is zero wherever the proposal histogram already bounds the main
 * weights (lossfun_outer, :85-102: clip(w - w_outer, min=0)) — on the benchmark configuration the 256-sample level
 * receives NO gradient at all during the first steps and the 96-sample level on 1-2 % of its samples
 * (profiles/r02_study_proposal_sparsity.txt). autograd still runs the whole chain on zeros
 * (ray_samplers.py:590-609 -> get_weights backward -> density field backward -> index_put). Here:
 *   nsamd_weights_bwd_gate      computes dL/d density as nsamd_weights_bwd does and RAISES *gate (a uint32 in device
 *                               memory, cleared on the stream by the call itself) when any ray carries gradient —
 *                               a non-zero or NaN upstream gradient, or a NaN / Inf / negative optical thickness;
 *   nsamd_density_mlp_bwd_gated, nsamd_hashgrid_encode_bwd_gated, nsamd_hashgrid_encode_bwd_rays_gated
 *                               return at once while *gate == 0: every value they would add is an exact zero
 *                               (finite parameters assumed), so the zero-filled gradients ARE the result. `denc` is
 *                               then not written and not read. With the gate raised they do what their ungated
 *                               forms do, bit for bit.
 * `ray_mask` (nullable; [num_rays] bytes written by nsamd_weights_bwd_gate's `ray_mask_out`: 1 = the ray carries gradient)
 * is the per-ray form of the flag for levels that are only PARTLY without gradient (measured on the benchmark run,
 * profiles/r03_proposal_sparsity.txt: a few per cent of the rays between the dense phases): the gated kernels neither load
 * nor write anything that belongs to a ray whose byte is 0 — density_mlp_bwd_gated skips the 256-point chunks without a
 * marked ray (their `denc` stays unwritten), the scatter's route pass and the ray-gradient kernel treat such samples as
 * the zeros they are. Ray mode only (pts.positions == NULL).
 * The gated scatter accumulates (the caller zero-fills dtable), needs the binned workspace and takes no dpositions. */
int nsamd_hashgrid_encode_bwd_gated(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                                    nsamd_grid grid, const float* denc, int64_t stride_p, int64_t stride_k,
                                    float* dtable, float* workspace, int64_t workspace_floats, const uint32_t* gate,
                                    const uint8_t* ray_mask, nsamd_stream_t stream);
int nsamd_hashgrid_encode_bwd_rays_gated(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb,
                                         const float* table, nsamd_grid grid, const float* denc, int64_t stride_p,
                                         int64_t stride_k, float* d_origins, float* d_directions, int accumulate,
                                         const uint32_t* gate, const uint8_t* ray_mask, nsamd_stream_t stream);

/* Words of scratch the binned scatter of nsamd_hashgrid_encode_bwd (write_only = 0) / nsamd_hashgrid_encode_bwd_set
 * (write_only = 1) wants for (grid, M); 0 when that path does not apply (M <= 0 or an unsupported grid). Host-only,
 * no device work. */
int64_t nsamd_hashgrid_encode_bwd_workspace(nsamd_grid grid, int64_t M, int write_only);

/* Leading words of that workspace which must be zero before its first use (header + per-tile cursors). Host-only. */
int64_t nsamd_hashgrid_encode_bwd_workspace_state(nsamd_grid grid, int64_t M);

/* Diagnostics of a scatter workspace since it was zeroed: events_host[0] = records that did not fit their tile's
 * queue and went through the spill list, [1] = of those, records applied with float atomics in no fixed order (more
 * than 8192 spills in one call: the result of that call was exact but not bit-reproducible), [2] = records lost
 * (must be 0). Synchronises `stream`. */
int nsamd_hashgrid_scatter_events(const float* workspace, uint32_t* events_host, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * SH encoding, 4 levels = 16 components (SHEncoding.pytorch_fwd, encodings.py:791-794 ->
 * utils/spherical_harmonics.py:24-93), evaluated on the input as given. dirs [M,3] -> out [M,16]. No gradient.
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_sh4_encode(const float* dirs, int64_t M, float* out, nsamd_stream_t stream);

/* NeRFEncoding.forward without covariances (field_components/encodings.py:148-189, vanilla-nerf's position / direction
 * encoding): out [M, 6 F (+3)] = [sin(s), sin(s + pi/2), x] with s[d F + f] = (2 pi x_d) freqs[f]; freqs = the device copy
 * of 2 ** torch.linspace(min_freq_exp, max_freq_exp, F) (host-evaluated, so the fp32 values are the reference's). The points
 * are [M,3] positions or rays + bin edges (sample midpoints, never materialised) as for the hash kernels. No gradient. */
int nsamd_nerf_encode(nsamd_points pts, int64_t M, const float* freqs, int32_t num_frequencies, int32_t include_input,
                      float* out, nsamd_stream_t stream);

/* SceneContraction(order=inf) forward on [M,3] (spatial_distortions.py:66-69). */
int nsamd_contract_linf(const float* x, int64_t M, float* out, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Proposal density head: MLP in(=2L) -> H (ReLU) -> 1, density = avg_init * trunc_exp(.) * selector
 * (HashMLPDensityField.get_density, fields/density_fields.py:104-117; MLP.pytorch_fwd mlp.py:160-179;
 * trunc_exp activations.py:28-54). enc is feature-major [in_dim, M]. Weights as nn.Linear: W0 [H,in], b0 [H],
 * W1 [1,H], b1 [1]. Supported: in_dim <= 32, H <= 64.
 * pre (nullable) receives the pre-activation (needed by the backward).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct nsamd_density_mlp {
  const float* W0; const float* b0; const float* W1; const float* b1;
  int32_t in_dim; int32_t hidden;
  float average_init_density;
} nsamd_density_mlp;

int nsamd_density_mlp_fwd(const float* enc, const float* selector, int64_t M, nsamd_density_mlp mlp,
                          float* density, float* pre, nsamd_stream_t stream);

/* The whole proposal-network forward in ONE kernel: transform -> hash grid (every level) -> MLP -> trunc_exp, one point
 * per lane, encoded features in registers (HashMLPDensityField.get_density, fields/density_fields.py:94-117 — what the
 * reference runs as tcnn.NetworkWithInputEncoding, field_components/mlp.py:252-269, on its tcnn path). enc (nullable,
 * feature-major [2L, M]), selector (nullable) and pre (nullable) are written only when given: a training step whose
 * proposal networks receive no gradient (ray_samplers.py:590) passes NULL and skips 4 * 2L bytes per point each way.
 * Bit-identical to nsamd_hashgrid_encode_fwd + nsamd_density_mlp_fwd. Built for (levels, hidden) in {5, 8} x {16, 64};
 * NSAMD_ERR_UNSUPPORTED otherwise (use the two-kernel pair). */
int nsamd_density_field_fwd(nsamd_points pts, int64_t M, int transform, nsamd_aabb aabb, const float* table,
                            nsamd_grid grid, nsamd_density_mlp mlp, float* enc, float* selector, float* density,
                            float* pre, nsamd_stream_t stream);

/* Backward: ddensity [M] -> denc feature-major [in_dim,M] (overwritten), and dW0,db0,dW1,db1 accumulated
 * (caller zero-fills). trunc_exp backward clamps the exponent to [-15,15] (activations.py:39-42).
 * workspace (nullable): >= 1024 * 1092 floats of scratch (one row of partial weight-gradient sums per workgroup,
 * summed in a fixed order: bit-reproducible). Without it the partial sums meet through float atomics (same values up to
 * summation order). One workspace serves one call at a time. */
int nsamd_density_mlp_bwd(const float* enc, const float* selector, const float* pre, const float* ddensity,
                          int64_t M, nsamd_density_mlp mlp, float* denc, float* dW0, float* db0, float* dW1,
                          float* db1, float* workspace, int64_t workspace_floats, nsamd_stream_t stream);

/* Gated form (see "zero-gradient gating" above): returns at once while *gate == 0, else identical to
 * nsamd_density_mlp_bwd (same workspace contract). */
int nsamd_density_mlp_bwd_gated(const float* enc, const float* selector, const float* pre, const float* ddensity,
                                int64_t M, nsamd_density_mlp mlp, float* denc, float* dW0, float* db0, float* dW1,
                                float* db1, float* workspace, int64_t workspace_floats, const uint32_t* gate,
                                const uint8_t* ray_mask, int32_t samples_per_ray, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * nerfacto main field head (NerfactoField.get_density + get_outputs, fields/nerfacto_field.py:203-310):
 *   base MLP 32 -> 64 (ReLU) -> 16 ; density = avg_init * trunc_exp(out[0]) * selector ; geo = out[1:16]
 *   head MLP [SH16(dir') | geo15 | appearance32] = 63 -> 64 -> 64 -> 3, sigmoid ; dir' = (dir+1)/2
 * fp32 MFMA (v_mfma_f32_16x16x4_f32), one wavefront per 16 samples, weights staged once per workgroup in LDS.
 * enc: feature-major [32, M]. directions: [num_dirs,3] with point p using row p / dir_group (dir_group = S for
 * per-ray directions, 1 for per-point). camera_indices likewise ([num_dirs] int64) — NULL selects
 * `appearance_const` ([32], the eval-time mean/zero embedding, nerfacto_field.py:253-261) for every point.
 * nsamd_field_mlp_fwd with rgb == NULL: the density alone (Field.density_fn, fields/base_field.py:64-77 — what the instant-ngp
 * sampler asks of its candidates, ray_samplers.py:420-429): the head MLP is not evaluated.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct nsamd_field_mlp {
  const float* base_W0; const float* base_b0;   /* [64,32],[64] */
  const float* base_W1; const float* base_b1;   /* [16,64],[16] */
  const float* head_W0; const float* head_b0;   /* [64,63],[64] */
  const float* head_W1; const float* head_b1;   /* [64,64],[64] */
  const float* head_W2; const float* head_b2;   /* [3,64],[3]   */
  const float* appearance;                      /* [num_images,32] embedding table (may be NULL if unused) */
  int32_t num_images;
  float average_init_density;
  /* Ray terms (nullable, both NULL = the plain kernels). 48 of head layer 0's 63 inputs — SH16 of the view direction and
   * the appearance row — are the same for every sample of a ray, so their share of the layer's pre-activation,
   *   ray_terms[ray] = head_b0 + head_W0[:, SH | appearance] . [SH16(dir') | appearance row]          ([num_rays,64])
   * is computed ONCE per ray (nsamd_field_ray_terms) and the per-sample GEMM of head layer 0 keeps K = 16 (the geo
   * features) instead of 64: 8 320 of a sample's 11 392 MACs instead of all of them, the same sums in another order
   * (nerfacto_field.py:283-310: `torch.cat([d, density_embedding, embedded_appearance])` into one Linear). Used by the
   * forward and backward entry points below whenever every 16-sample tile lies inside one ray (dir_group % 16 == 0,
   * M % dir_group == 0); the backward additionally needs `ray_inputs` ([num_rays,48] = SH16 | appearance32, written by the
   * same call) for the weight gradient of those 48 columns and a workspace with room for 64 floats per tile, and falls
   * back to the plain kernel otherwise. The caller recomputes the terms whenever head_W0 / head_b0 / the appearance
   * table / the directions / the camera indices changed. */
  const float* ray_terms;
  const float* ray_inputs;
} nsamd_field_mlp;

/* ray_terms [num_rays,64] (and ray_inputs [num_rays,48], nullable) of the struct above for `num_rays` rays: directions
 * [num_rays,3]; camera_indices [num_rays] int64 or NULL (then appearance_const [32], or neither: no appearance embedding).
 * `mlp.ray_terms / ray_inputs` are ignored here. */
int nsamd_field_ray_terms(const float* directions, const int64_t* camera_indices, const float* appearance_const,
                          int64_t num_rays, nsamd_field_mlp mlp, float* ray_terms, float* ray_inputs,
                          nsamd_stream_t stream);

int nsamd_field_mlp_fwd(const float* enc, const float* selector, const float* directions,
                        const int64_t* camera_indices, const float* appearance_const, int64_t dir_group, int64_t M,
                        nsamd_field_mlp mlp, float* density, float* rgb, nsamd_stream_t stream);

typedef struct nsamd_field_mlp_grads {          /* all accumulated; caller zero-fills */
  float* base_W0; float* base_b0; float* base_W1; float* base_b1;
  float* head_W0; float* head_b0; float* head_W1; float* head_b1; float* head_W2; float* head_b2;
  float* appearance;                            /* [num_images,32] or NULL */
} nsamd_field_mlp_grads;

/* Backward recomputes the activations from enc (nothing but enc is kept from the forward).
 * ddensity [M], drgb [M,3] -> denc feature-major [32,M] (overwritten) + parameter gradients.
 * workspace (nullable): >= 256 CUs * 12544 floats of device scratch for per-workgroup weight-gradient partials
 * (summed by a single-writer pass); without it the workgroups flush with global atomics. */
int nsamd_field_mlp_bwd(const float* enc, const float* selector, const float* directions,
                        const int64_t* camera_indices, const float* appearance_const, int64_t dir_group, int64_t M,
                        nsamd_field_mlp mlp, const float* ddensity, const float* drgb, float* denc,
                        nsamd_field_mlp_grads grads, float* workspace, int64_t workspace_floats,
                        nsamd_stream_t stream);

/* nsamd_field_mlp_bwd + nsamd_hashgrid_encode_bwd_set in one: the backward of the main field INCLUDING its hash table's
 * gradient (the backward of HashEncoding.pytorch_fwd, field_components/encodings.py:417-458, of the features this field
 * was evaluated on). The data gradient of base layer 0 is the encoded-feature gradient, and the persistent workgroups of
 * the MLP kernel emit the scatter's pass-1 records from their registers — one static queue segment per (workgroup, table
 * tile), slot = LDS rank — instead of storing `denc`, launching the route pass and having it load the gradients again and
 * recompute every cell. The order-independent fixed-point apply pass of the scatter follows unchanged, so the table
 * gradient is what nsamd_hashgrid_encode_bwd_set(denc) writes up to the position of the fixed-point truncation
 * (deterministic, run-to-run bit-identical; `dtable` [L*T,2] is WRITTEN, no zero-fill needed).
 * pts / transform / aabb / grid: the points and grid `enc` was encoded with (nsamd_hashgrid_encode_fwd); 16 levels.
 * denc: NULL, or feature-major [32,M] when the caller wants the feature gradient as well (camera optimiser).
 * scatter_workspace: nsamd_field_mlp_bwd_scatter_workspace(grid, M, &state) floats, the first `state` of them zero
 * before the first call (the kernels leave them zero). Returns NSAMD_ERR_UNSUPPORTED for other level counts. */
int64_t nsamd_field_mlp_bwd_scatter_workspace(nsamd_grid grid, int64_t M, int64_t* state_words);
/* Leave `cus` compute units out of the persistent workgroups of the NEXT launches of the field backward (all four entry
 * points; 0 = none, the default; -1 = as many as ONE more sweep over the tiles frees — the workgroups take
 * ceil(tile groups / workgroups) sweeps whatever their number, so that is the cheapest reservation: 6 sweeps on 256 CUs
 * become 7 on 220 for 196 608 points) and return the previous setting. The backward's workgroups own a CU's LDS and registers
 * for the whole launch; on the iterations whose proposal networks receive gradient (model_components/ray_samplers.py:
 * 590-609) their latency-bound backward chains, queued on another stream, otherwise wait for its end. Process-wide,
 * read when a launch is issued (or captured). Results are the same gradients summed in another fixed order. */
int nsamd_field_mlp_bwd_reserve_cus(int cus);
int nsamd_field_mlp_bwd_scatter(nsamd_points pts, int transform, nsamd_aabb aabb, nsamd_grid grid, const float* enc,
                                const float* selector, const float* directions, const int64_t* camera_indices,
                                const float* appearance_const, int64_t dir_group, int64_t M, nsamd_field_mlp mlp,
                                const float* ddensity, const float* drgb, float* denc, nsamd_field_mlp_grads grads,
                                float* workspace, int64_t workspace_floats, float* dtable, float* scatter_workspace,
                                int64_t scatter_workspace_floats, nsamd_stream_t stream);
/* The same call one launch group at a time (per-kernel timing of the benchmark's roofline leg; the groups in order give the
 * single call's bits): phase 1 = the gradient kernel with the record emission, 2 = the weight-gradient reduce, 4 = the
 * scatter's apply + finish passes over the records phase 1 left in the queues; 6 = 2 and 4 as the single call issues them (the
 * reduce riding the apply pass) — for a caller that puts work of its own between the gradient kernel and the rest. */
int nsamd_field_mlp_bwd_scatter_phase(nsamd_points pts, int transform, nsamd_aabb aabb, nsamd_grid grid, const float* enc,
                                      const float* selector, const float* directions, const int64_t* camera_indices,
                                      const float* appearance_const, int64_t dir_group, int64_t M, nsamd_field_mlp mlp,
                                      const float* ddensity, const float* drgb, float* denc, nsamd_field_mlp_grads grads,
                                      float* workspace, int64_t workspace_floats, float* dtable, float* scatter_workspace,
                                      int64_t scatter_workspace_floats, int phase, nsamd_stream_t stream);

/* nsamd_field_mlp_bwd in two launches, so that a caller can put the second on another stream: phase 1 = the gradient
 * kernel (denc + the per-workgroup weight-gradient partials in `workspace`, which is REQUIRED here), phase 2 = the
 * fixed-order sum of the partials into `grads` (needs nothing but workspace, grads, camera_indices and the sizes). Phase 2
 * depends on phase 1 only; the table scatter that consumes denc (nsamd_hashgrid_encode_bwd) does not depend on phase 2
 * (MLPWithHashEncoding backward, field_components/mlp.py:187-295: weight gradients and input gradients are independent
 * products of the same upstream gradient). Same bits as the single call. */
int nsamd_field_mlp_bwd_phase(const float* enc, const float* selector, const float* directions,
                              const int64_t* camera_indices, const float* appearance_const, int64_t dir_group, int64_t M,
                              nsamd_field_mlp mlp, const float* ddensity, const float* drgb, float* denc,
                              nsamd_field_mlp_grads grads, float* workspace, int64_t workspace_floats, int phase,
                              nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Generic dense layer for the stand-alone MLP of the plugin API (MLP.pytorch_fwd, field_components/mlp.py:160-179):
 * y[M,N] = act(x[M,K] W[N,K]^T + b[N]); activation 0 = none, 1 = ReLU, 2 = Sigmoid, 3 = Softplus (the DensityFieldHead of
 * vanilla-nerf, field_heads.py:98-108); any K, N (layers wider than 128 run as 128 x 128 blocks of W: the 8 x 256 MLP with
 * its 319-wide skip layer, mlp.py:143-158); fp32 MFMA.
 * Backward: dx[M,K] (nullable, overwritten), dW[N,K] / db[N] accumulated (nullable); y = the forward's output.
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_linear_fwd(const float* x, const float* W, const float* b, int64_t M, int32_t K, int32_t N, int activation,
                     float* y, nsamd_stream_t stream);
int nsamd_linear_bwd(const float* x, const float* W, const float* y, const float* dy, int64_t M, int32_t K, int32_t N,
                     int activation, float* dx, float* dW, float* db, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Samplers (model_components/ray_samplers.py). Bins are [num_rays, S+1]; `s` = normalised spacing domain,
 * `t` = euclidean distance. lin_host-free: `edges` is the device copy of torch.linspace(0,1,S+1) and `u_base` of
 * torch.linspace(0, 1-1/(S+1), S+1) (the host evaluates them with torch so the fp32 values are the reference's).
 * jitter (nullable = eval) is the raw U[0,1) draw: one per ray ([num_rays], single_jitter=True, jitter_per_edge = 0) or
 * one per bin edge ([num_rays, S+1], single_jitter=False, jitter_per_edge = 1; ray_samplers.py:104-107, 318-322).
 * ------------------------------------------------------------------------------------------------------------ */

/* SpacedSampler.generate_ray_samples (ray_samplers.py:78-128): spacing 0 = UniformLinDispPiecewiseSampler
 * (:225-248, nerfacto default), 1 = UniformSampler (:131-155, the Blender benchmark recipe). */
int nsamd_piecewise_bins(const float* nears, const float* fars, const float* edges, const float* jitter,
                         int32_t jitter_per_edge, int64_t num_rays, int32_t S, int spacing, float* s_bins,
                         float* t_bins, nsamd_stream_t stream);

/* RaySamples.get_weights (cameras/rays.py:129-152): left-to-right cumsum per ray. */
int nsamd_weights_fwd(const float* t_bins, const float* density, int64_t num_rays, int32_t S, float* weights,
                      nsamd_stream_t stream);
int nsamd_weights_bwd(const float* t_bins, const float* density, const float* dweights, int64_t num_rays,
                      int32_t S, float* ddensity, nsamd_stream_t stream);
/* The same, and *gate_out = (any ray carries gradient) — see "zero-gradient gating" under the hash encoding.
 * gate_precleared != 0: the caller has zeroed *gate_out on the stream since its last use (a training schedule clears all its
 * flags in one fill off the critical path instead of one 4-byte memset node per level); 0: the call clears it itself. */
int nsamd_weights_bwd_gate(const float* t_bins, const float* density, const float* dweights, int64_t num_rays,
                           int32_t S, float* ddensity, uint32_t* gate_out, uint8_t* ray_mask_out, int32_t gate_precleared,
                           nsamd_stream_t stream);

/* ---- backward of ALL proposal levels of an update iteration, stage by stage across the levels -------------------------
 * What ProposalNetworkSampler's levels receive through interlevel_loss on the steps that update them
 * (model_components/ray_samplers.py:590-609): per level the gated chain
 *   nsamd_weights_bwd_gate -> nsamd_density_mlp_bwd_gated -> nsamd_hashgrid_encode_bwd_gated      (ray mode, stride 1 / M)
 * The levels share nothing (own network, table, gradients, scratch) and each of the chain's six launches is as long as
 * its slowest workgroup's memory round trips, not as its work; here the SAME stage of two levels is ONE launch (level
 * pairs (0,1), (2,3), ...; an odd level out, levels that share a workspace, network or table gradient, or shapes the
 * merged kernels do not cover go through the per-level entry points inside the call). Results: bit for bit those of the
 * per-level calls. One struct per level: */
typedef struct {
  int64_t num_rays;
  int32_t samples_per_ray;
  /* RaySamples.get_weights backward: bins [num_rays, S+1], density / dweights [num_rays, S] -> ddensity; *gate and
   * ray_mask [num_rays] are WRITTEN (see "zero-gradient gating") */
  const float* t_bins;
  const float* density;
  const float* dweights;
  float* ddensity;
  uint32_t* gate;
  uint8_t* ray_mask;
  /* density MLP backward (as nsamd_density_mlp_bwd_gated): enc [in_dim, M], selector (nullable), pre [M] -> denc [in_dim, M];
   * dW0 .. db1 accumulate; mlp_workspace as there */
  const float* enc;
  const float* selector;
  const float* pre;
  nsamd_density_mlp mlp;
  float* denc;
  float* dW0;
  float* db0;
  float* dW1;
  float* db1;
  float* mlp_workspace;
  int64_t mlp_workspace_floats;
  /* table scatter (as nsamd_hashgrid_encode_bwd_gated on ray-mode points origins / directions / t_bins): dtable accumulates */
  const float* origins;
  const float* directions;
  int transform;
  nsamd_aabb aabb;
  const float* table;
  nsamd_grid grid;
  float* dtable;
  float* scatter_workspace;
  int64_t scatter_workspace_floats;
} nsamd_proposal_level_bwd;
/* gates_precleared as in nsamd_weights_bwd_gate (0: every level's *gate is cleared on the stream first). */
int nsamd_proposal_levels_bwd(const nsamd_proposal_level_bwd* levels, int32_t num_levels, int32_t gates_precleared,
                              nsamd_stream_t stream);

/* PDFSampler.generate_ray_samples (ray_samplers.py:276-372) preceded by the anneal pow(weights, anneal)
 * (ray_samplers.py:601; skipped when anneal == 1). include_original = 0: s_bins / t_bins are [num_rays, S+1] (the
 * proposal sampler); 1: the new edges merged with the existing ones and sorted (ray_samplers.py:356-357, vanilla-nerf's
 * fine sampler) -> [num_rays, S_prev + S + 2]. inds (nullable) receives the
 * searchsorted(side="right") result as int32 [num_rays, S+1]. u_offset = (float)(1.0 / (2 * (S+1))) is the eval-mode
 * offset (ray_samplers.py:327), rounded double->float by the host like torch rounds the Python scalar.
 * INDEX CONTRACT (north_star: "bit-exact for sample indices"). The only sum of this stage whose order the reference does
 * not fix is weights_sum = torch.sum(weights, dim=-1) (ray_samplers.py:304): ATen's CPU kernel reduces a row with a
 * vectorised cascade whose grouping depends on the host's SIMD width, its CUDA kernel with a block tree — the reference
 * itself returns sums that differ in the last ulp between its own backends. This library evaluates the CORRECTLY ROUNDED
 * sum (accumulated in double, rounded to fp32 once) — the value every order approximates — and torch.cumsum's order for the
 * CDF (left to right, double accumulate, per-element rounding: ATen's CPU cumsum). Consequence, checked against the
 * fixtures the reference wrote (tests/golden/samplers.npz, _check_inds in tests/test_oracle_vs_golden.py): all indices
 * equal the reference's EXCEPT where the sample position u coincides with a CDF entry to within 2 ulp — an exact tie that
 * the last bit of weights_sum decides (on the fixtures: <= 4 of 3 104 indices, the eval-mode u = 0.5 of degenerate rays
 * whose CDF hits 0.5). TIE RULE: at such a tie the index follows searchsorted(side="right") on THIS library's cdf, i.e.
 * the first edge strictly greater than u; the two candidate indices bracket the same CDF value, so the interpolated bin
 * (the sample position the caller uses) is the same to 2 ulp either way. Against the oracle (same rounding) indices and
 * bins are bit-exact on every test seed.
 * anneal_dev (nullable): device copy of the anneal exponent; overrides `anneal` so that a captured hipGraph of the
 * training step can be replayed while the schedule advances. spacing: as in nsamd_piecewise_bins (the s -> t map of
 * the initial sampler, ray_samplers.py:112-116). */
int nsamd_pdf_resample(const float* s_bins_prev, const float* weights, int32_t S_prev, const float* u_base,
                       const float* jitter, const float* nears, const float* fars, float anneal,
                       const float* anneal_dev, float histogram_padding, float eps, float u_offset, int spacing,
                       int32_t jitter_per_edge, int32_t include_original, int64_t num_rays, int32_t S, float* s_bins,
                       float* t_bins, int32_t* inds, nsamd_stream_t stream);

/* One proposal level of ProposalNetworkSampler.generate_ray_samples (ray_samplers.py:576-617) in a single launch:
 * weights = RaySamples.get_weights(density) of the level's samples (t_bins_prev), its median depth (nullable;
 * models/nerfacto.py:346-347 renders prop_depth_i for every level), then nsamd_pdf_resample on those weights.
 * Same numbers as nsamd_weights_fwd + nsamd_composite_fwd(median) + nsamd_pdf_resample. */
int nsamd_proposal_resample(const float* t_bins_prev, const float* s_bins_prev, const float* density, int32_t S_prev,
                            const float* u_base, const float* jitter, const float* nears, const float* fars,
                            float anneal, const float* anneal_dev, float histogram_padding, float eps, float u_offset,
                            int spacing, int64_t num_rays, int32_t S, float* weights, float* depth_median,
                            float* s_bins, float* t_bins, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Compositing (model_components/renderers.py): RGBRenderer.combine_rgb :72-119 (+ eval nan_to_num/clamp
 * :225-231), AccumulationRenderer :293-317, DepthRenderer median :354-364 and expected :365-383.
 * background: 0 = "random"/none, 1 = "last_sample", 2 = constant colour bg_rgb[3] (host values).
 * Outputs (any may be NULL): rgb_out [N,3], acc [N], depth_expected [N] (clipped to the batch-global min/max of
 * the sample midpoints, as the reference does), depth_median [N], median_idx [N] int32.
 * `workspace` = 2 + 2*ceil(num_rays/4) floats of device scratch (final and per-workgroup min / max of the sample
 * midpoints), only needed when depth_expected != NULL; words 0..1 are read again by the backward.
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_composite_fwd(const float* rgb, const float* weights, const float* t_bins, int64_t num_rays, int32_t S,
                        int background, const float* bg_rgb_host, int eval_mode, float* rgb_out, float* acc,
                        float* depth_expected, float* depth_median, int32_t* median_idx, float* workspace,
                        nsamd_stream_t stream);

/* Backward of rgb_out / acc / depth_expected w.r.t. rgb samples and weights (training mode).
 * d_rgb_out [N,3], d_acc [N] (nullable), d_depth [N] (nullable; needs `workspace` from the forward and t_bins).
 * d_weights_add (nullable) [N,S] is added to d_weights (gradient of another loss on the same weights). */
int nsamd_composite_bwd(const float* rgb, const float* weights, const float* t_bins, int64_t num_rays, int32_t S,
                        int background, const float* bg_rgb_host, const float* d_rgb_out, const float* d_acc,
                        const float* d_depth, const float* workspace, const float* d_weights_add, float* d_rgb,
                        float* d_weights, nsamd_stream_t stream);

/* Training-step fusions of the above (same numbers, fewer launches):
 * nsamd_render_train = nsamd_weights_fwd (weights [N,S] out) + nsamd_composite_fwd (training mode) + the MSE loss of the
 * composited colour against target [N,3]: sq_err [N] (nullable) = per-ray sum of squared errors, d_rgb_out [N,3]
 * (nullable) = 2 (rgb_out - target) grad_scale. target may be NULL (no loss).
 * nsamd_render_train_bwd = nsamd_composite_bwd (d_rgb_out, d_weights_add) + nsamd_weights_bwd: d_rgb [N,S,3] and
 * d_density [N,S]; `weights` are the forward's.
 * background = 3 (these two only): background_color="random" in training — rgb_out is the composite without a background
 * (renderers.py:112-115) and the loss is taken on rgb_out + bg_rays[ray] (1 - acc), bg_rays [N,3] = the caller's
 * rand_like(pred) draw (blend_background_for_loss_computation, renderers.py:194-196). bg_rays is NULL otherwise. */
int nsamd_render_train(const float* rgb, const float* density, const float* t_bins, int64_t num_rays, int32_t S,
                       int background, const float* bg_rgb_host, const float* target, float grad_scale, float* weights,
                       float* rgb_out, float* acc, float* depth_expected, float* depth_median, float* workspace,
                       float* sq_err, float* d_rgb_out, const float* bg_rays, nsamd_stream_t stream);
int nsamd_render_train_bwd(const float* rgb, const float* weights, const float* density, const float* t_bins,
                           int64_t num_rays, int32_t S, int background, const float* bg_rgb_host,
                           const float* d_rgb_out, const float* d_weights_add, float* d_rgb, float* d_density,
                           const float* bg_rays, nsamd_stream_t stream);

/* scale_gradients_by_distance_squared (model_components/losses.py:534-569; NerfactoModelConfig.use_gradient_scaling,
 * models/nerfacto.py:321-322): the gradients reaching the field's per-sample outputs are multiplied by
 * clamp(((t_start + t_end) / 2)^2, 0, 1), in place. d_density [N,S] and d_rgb [N,S,3] nullable. */
int nsamd_distance_gradient_scale(const float* t_bins, int64_t num_rays, int32_t S, float* d_density, float* d_rgb,
                                  nsamd_stream_t stream);

/* MSELoss (model_components/losses.py:31): loss_sum += sum((pred-target)^2) (caller zeroes; mean = /n),
 * dpred (nullable) = 2 (pred-target) grad_scale  with grad_scale = upstream / n. */
int nsamd_mse_loss(const float* pred, const float* target, int64_t n, float grad_scale, float* loss_sum, float* dpred,
                   nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Proposal losses (model_components/losses.py). Per-ray fused forward + gradient:
 * interlevel (:53-131): per_ray_loss[n] = sum_i clip(w_i - outer_i, 0)^2 / (w_i + 1e-7); dwp = d(sum)/d(wp).
 * distortion (:135-154): per_ray_loss[n] and dw. The host applies mean() and the loss multipliers to the value;
 * the gradients are pre-multiplied by grad_scale (= multiplier / number of terms in the mean). dw may be NULL.
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_interlevel_loss(const float* s_bins_fine, const float* w_fine, int32_t S_fine, const float* s_bins_prop,
                          const float* w_prop, int32_t S_prop, int64_t num_rays, float grad_scale,
                          float* per_ray_loss, float* dw_prop, nsamd_stream_t stream);
int nsamd_distortion_loss(const float* s_bins, const float* weights, int32_t S, int64_t num_rays, float grad_scale,
                          float* per_ray_loss, float* dweights, nsamd_stream_t stream);

/* All proposal losses of a training step in one launch (models/nerfacto.py:367-375): the interlevel loss of each of
 * `levels` (<= 4) proposal levels against the fine samples, and the distortion loss of the fine samples. The pointer
 * arrays are HOST arrays of `levels` device pointers; dw_prop (or any entry) and dw_distortion may be NULL. */
int nsamd_proposal_losses(const float* s_bins_fine, const float* w_fine, int32_t S_fine, int32_t levels,
                          const float* const* s_bins_prop, const float* const* w_prop, const int32_t* S_prop,
                          int64_t num_rays, float interlevel_grad_scale, float distortion_grad_scale,
                          float* const* interlevel_per_ray, float* const* dw_prop, float* distortion_per_ray,
                          float* dw_distortion, nsamd_stream_t stream);

/* The iteration's loss values and training metrics from the per-ray terms the launches above left behind (nsamd_render_train's
 * sq_err, nsamd_proposal_losses' per-ray values): one small launch instead of a dozen host-issued reductions for a trainer that
 * logs the loss dictionary every iteration (engine/trainer.py:487-531). loss_values: 32 floats = 8 results + scratch of the
 * pass (partial sums and a ticket word that must be ZERO before the first launch and resets itself); the values as
 * models/nerfacto.py:363-375 scales them and the two training metrics of :352-361, summed in a fixed order in double:
 * [0] rgb_loss = sum(sq_err) / (3 N), [1] interlevel_loss = interlevel_loss_mult * sum over levels and rays / (N S),
 * [2] distortion_loss = distortion_loss_mult * sum(distortion_per_ray) / N, [3] psnr = -10 log10(rgb_loss),
 * [4] distortion = sum(distortion_per_ray) / N, [5] = [0] + [1] + [2]. */
int nsamd_train_loss_values(const float* sq_err, const float* distortion_per_ray, int32_t levels,
                            const float* const* interlevel_per_ray, int64_t num_rays, int32_t S, float interlevel_loss_mult,
                            float distortion_loss_mult, float* loss_values, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Packed-sample path of instant-ngp (BASELINE configs[3]): what the reference gets from nerfacc 0.5.2
 * (OccGridEstimator.sampling, pack_info, render_weight_from_density, render_visibility_from_density,
 * accumulate_along_rays; call sites model_components/ray_samplers.py:481-493, models/instant_ngp.py:192-198,
 * model_components/renderers.py:93-102, 310-314, 369-377). nerfacc is not part of /root/reference: the arithmetic is
 * restated (oracle/packed_oracle.py); the weight / visibility / accumulation formulas are anchored on the reference's
 * dense path, the marcher's sample placement is NOT pinned.
 * Samples of a ray are contiguous, rays in increasing order; packed_info [N,2] int64 = (start, count) per ray.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct nsamd_occgrid {
  const uint8_t* binaries; /* [levels, R, R, R] 0/1: level l covers the region of interest scaled by 2^l about its centre */
  int32_t levels;
  int32_t resolution;
  float aabb[6];           /* region of interest: min xyz, max xyz */
  const uint32_t* coarse;  /* nullable: one bit per 4x4x4 block of cells and level (nsamd_occgrid_binarise writes it;
                              nsamd_occgrid_coarse_words(levels, resolution) words, bit b of word w = block 32 w + b in
                              [level, bx, by, bz] order). The marcher stages it in LDS and skips the byte grid for steps
                              whose block is empty — same samples, by construction. */
} nsamd_occgrid;

/* Ray marching through the occupancy grid, two calls: _count fills counts[N] (int32), nsamd_packed_info turns them into
 * packed_info + the total, _write emits ray_indices [n] int64, t_starts / t_ends [n]. A ray marches t = t0, t0 + dt, ...
 * inside [max(near, t_min), min(far, t_max)] clipped to the outermost grid level, dt = clamp(t * cone_angle, step, 1e10);
 * a step is kept when the cell (finest level containing the step's midpoint) is occupied. jitter [N] in [0,1) (nullable)
 * shifts a ray's lattice by jitter * step (stratified training, ray_samplers.py:489). t_min / t_max nullable. */
int nsamd_occgrid_march_count(const float* origins, const float* directions, const float* t_min, const float* t_max,
                              int64_t num_rays, float near_plane, float far_plane, nsamd_occgrid grid, float step_size,
                              float cone_angle, const float* jitter, int32_t* counts, nsamd_stream_t stream);
int nsamd_occgrid_march_write(const float* origins, const float* directions, const float* t_min, const float* t_max,
                              int64_t num_rays, float near_plane, float far_plane, nsamd_occgrid grid, float step_size,
                              float cone_angle, const float* jitter, const int64_t* packed_info, int64_t* ray_indices,
                              float* t_starts, float* t_ends, nsamd_stream_t stream);
/* The same two calls marching every ray ONCE: _count_stash also leaves a ray's first stash_cap kept steps as (t_start, t_end)
 * pairs in stash [num_rays, stash_cap, 2]; _write_stashed copies them to their packed places and marches only the rays that
 * kept more than stash_cap steps a second time. Same counts, same values as the plain pair (stash == NULL: the plain pair). */
int nsamd_occgrid_march_count_stash(const float* origins, const float* directions, const float* t_min, const float* t_max,
                                    int64_t num_rays, float near_plane, float far_plane, nsamd_occgrid grid, float step_size,
                                    float cone_angle, const float* jitter, int32_t* counts, float* stash, int32_t stash_cap,
                                    nsamd_stream_t stream);
int nsamd_occgrid_march_write_stashed(const float* origins, const float* directions, const float* t_min, const float* t_max,
                                      int64_t num_rays, float near_plane, float far_plane, nsamd_occgrid grid,
                                      float step_size, float cone_angle, const float* jitter, const int64_t* packed_info,
                                      const float* stash, int32_t stash_cap, int64_t* ray_indices, float* t_starts,
                                      float* t_ends, nsamd_stream_t stream);

/* Occupancy-grid maintenance — what nerfacc's OccGridEstimator.update_every_n_steps does around `occ_eval_fn` (call site
 * models/instant_ngp.py:151-156; nerfacc 0.5.2 restated, see above):
 *   _cell_positions   positions [M,3] inside the cells `cells` [M] (flat index over [levels, R, R, R]; NULL: cell i = i) at
 *                     the fractional offsets jitter [M,3] in [0,1) — the points the density is evaluated at;
 *   _update           occs[c] = max(occs[c] * ema_decay, the new estimates of cell c)  for the listed cells (repeats
 *                     allowed, the result does not depend on the thread order); scratch: total_cells floats;
 *   _binarise         binaries = occs > min(mean(occs), occ_thre) (mean in double, fixed summation order) and the coarse
 *                     bitfield of the result (nullable); scratch: 1024 doubles; threshold_out (nullable) [2] = (threshold,
 *                     mean) in device memory. */
int64_t nsamd_occgrid_coarse_words(int32_t levels, int32_t resolution);
int nsamd_occgrid_cell_positions(const int64_t* cells, int64_t M, nsamd_occgrid grid, const float* jitter, float* positions,
                                 nsamd_stream_t stream);
int nsamd_occgrid_update(float* occs, int64_t total_cells, const int64_t* cells, const float* occ_new, int64_t M,
                         float ema_decay, float* scratch, nsamd_stream_t stream);
int nsamd_occgrid_binarise(const float* occs, int32_t levels, int32_t resolution, float occ_thre, uint8_t* binaries,
                           uint32_t* coarse, double* scratch, float* threshold_out, nsamd_stream_t stream);

/* nerfacc.pack_info from per-ray counts: packed_info [N,2] int64 and total[0] (device int64) = number of samples. */
int nsamd_packed_info(const int32_t* counts, int64_t num_rays, int64_t* packed_info, int64_t* total, nsamd_stream_t stream);

/* nerfacc.render_weight_from_density: w = T (1 - exp(-sigma dt)), T = exp(-sum of sigma dt in front) — one wavefront per
 * ray scans its segment (double running sum). transmittance nullable. Backward: dL/dweights -> dL/dsigmas. */
int nsamd_packed_weights_fwd(const float* t_starts, const float* t_ends, const float* sigmas, const int64_t* packed_info,
                             int64_t num_rays, float* weights, float* transmittance, nsamd_stream_t stream);
int nsamd_packed_weights_bwd(const float* t_starts, const float* t_ends, const float* sigmas, const float* dweights,
                             const int64_t* packed_info, int64_t num_rays, float* dsigmas, nsamd_stream_t stream);

/* nerfacc.render_visibility_from_density: mask[s] = T_s >= early_stop_eps && alpha_s >= alpha_thre (uint8) and the
 * number of kept samples per ray — the same scan, stopping a ray at the first chunk whose transmittance is below the
 * threshold (visibility-ordered early termination). nsamd_packed_compact then moves the survivors (order kept) to the
 * layout given by packed_info_new (from the kept counts through nsamd_packed_info). */
int nsamd_packed_visibility(const float* t_starts, const float* t_ends, const float* sigmas, const int64_t* packed_info,
                            int64_t num_rays, float early_stop_eps, float alpha_thre, uint8_t* mask, int32_t* kept_counts,
                            nsamd_stream_t stream);
int nsamd_packed_compact(const uint8_t* mask, const int64_t* packed_info_old, const int64_t* packed_info_new,
                         int64_t num_rays, const float* t_starts, const float* t_ends, int64_t* ray_indices_out,
                         float* t_starts_out, float* t_ends_out, nsamd_stream_t stream);

/* Packed branches of RGBRenderer / AccumulationRenderer / DepthRenderer("expected") (accumulate_along_rays):
 * rgb [n,3], weights [n] -> out_rgb [N,3], accumulation [N], depth [N] (nullable; = sum w mid / (acc + 1e-10), the
 * caller clips it to the batch's midpoint range as renderers.py:381-383). background_mode 0: none ("random": blended in
 * the loss), 1: the host colour background_rgb_host[3] times (1 - accumulation). eval_mode: nan_to_num + clamp.
 * Backward: g_rgb [N,3], g_accumulation [N] (nullable) -> d_rgb [n,3] (nullable), d_weights [n]. */
int nsamd_packed_composite_fwd(const float* rgb, const float* weights, const float* t_starts, const float* t_ends,
                               const int64_t* packed_info, int64_t num_rays, int background_mode,
                               const float* background_rgb_host, int eval_mode, float* out_rgb, float* out_accumulation,
                               float* out_depth, nsamd_stream_t stream);
int nsamd_packed_composite_bwd(const float* rgb, const float* weights, const int64_t* ray_indices, int64_t num_samples,
                               int background_mode, const float* background_rgb_host, const float* g_rgb,
                               const float* g_accumulation, float* d_rgb, float* d_weights, nsamd_stream_t stream);

/* positions [n,3] of packed samples: o[ray] + d[ray] (t_start + t_end) / 2 (the sigma_fn of VolumetricSampler,
 * ray_samplers.py:420-429). */
int nsamd_packed_positions(const float* origins, const float* directions, const int64_t* ray_indices, const float* t_starts,
                           const float* t_ends, int64_t num_samples, float* positions, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Pinhole ray generation (RayGenerator.forward, model_components/ray_generators.py:41-56 ->
 * Cameras._generate_rays_from_coords perspective branch, cameras/cameras.py:598-634, 655-656, 781-787, 887-909).
 * ray_indices [N,3] int64 (camera,row,col); c2w [C,3,4]; fx,fy,cx,cy [C]. Pixel centres at +0.5.
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_raygen_pinhole(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy,
                         const float* cx, const float* cy, int64_t num_rays, int32_t num_cameras, float* origins,
                         float* directions, float* pixel_area, float* directions_norm, nsamd_stream_t stream);

/* The same rays for ONE camera's image in row-major pixel order, chunk by chunk: ray i = pixel first_pixel + i of the implicit
 * (row, col) grid of width `width` — what Model.get_outputs_for_camera (models/base_model.py:166-175) gets from
 * camera.generate_rays(camera_indices=0, keep_shape=True) and then slices into chunks (:178-205), generated straight into the
 * render loop's input buffers: no [H,W,3] bundle and no index list in HBM. c2w [3,4] (device), intrinsics by value; rays
 * num_rays .. padded_rays - 1 repeat the last pixel (the padding of a last, shorter chunk). Same bits as nsamd_raygen_pinhole. */
int nsamd_raygen_pinhole_grid(const float* c2w, float fx, float fy, float cx, float cy, int32_t width, int64_t first_pixel,
                              int64_t num_rays, int64_t padded_rays, float* origins, float* directions, float* pixel_area,
                              nsamd_stream_t stream);

/* Data-parallel exchange of a hash-table gradient whose coarse levels reach only a few of their rows (the torch path
 * hashes every level, encodings.py:398-415: level l touches at most (res_l + 1)^3 of its 2^log2_T rows): pack the
 * `n` reachable rows `index` (int64, sorted) of `rows` [*, feat] into `packed` [n, feat] before the all-reduce
 * (gather), unpack after it (scatter). Replaces DDP's dense bucket for that part of the table
 * (pipelines/base_pipeline.py:279-282). */
int nsamd_rows_gather(const float* rows, const int64_t* index, int64_t n, int32_t feat, float* packed,
                      nsamd_stream_t stream);
int nsamd_rows_scatter(float* rows, const int64_t* index, int64_t n, int32_t feat, const float* packed,
                       nsamd_stream_t stream);

/* The step's ray batch out of `slots` pre-generated batches resident in HBM (what VanillaDataManager.next_train hands
 * the model each iteration, data/datamanagers/base_datamanager.py:506-515): pools [slots, N, 3] fp32 (origins,
 * directions, target rgb) and [slots, N] int64 (camera indices) -> the [N,3] / [N] buffers of the step. The slot index
 * is read from DEVICE memory (slot_dev[0], a float like the other per-step scalars; clamped to [0, slots)), so a captured
 * hipGraph replays with a new batch every step. */
int nsamd_select_batch(const float* slot_dev, int32_t slots, int64_t num_rays, const float* origins_pool,
                       const float* directions_pool, const int64_t* cameras_pool, const float* target_pool,
                       float* origins, float* directions, int64_t* cameras, float* target, nsamd_stream_t stream);

/* nsamd_select_batch + nsamd_piecewise_bins in one launch (the head of a training iteration over a pool of batches: the
 * hand-over of base_datamanager.py:506-515 and the initial sampler of ProposalNetworkSampler, ray_samplers.py:78-128, 586).
 * Same numbers as the two launches; arguments as theirs. */
int nsamd_select_bins(const float* slot_dev, int32_t slots, int64_t num_rays, const float* origins_pool,
                      const float* directions_pool, const int64_t* cameras_pool, const float* target_pool, float* origins,
                      float* directions, int64_t* cameras, float* target, const float* nears, const float* fars,
                      const float* edges, const float* jitter, int32_t jitter_per_edge, int32_t S, int spacing,
                      float* s_bins, float* t_bins, nsamd_stream_t stream);

/* Head of a captured training iteration: what changes from step to step, produced on the device (a replayed hipGraph then
 * needs no host-issued upload in front of it). counter [2] int64 (device): [0] = row counter of `table`, [1] = draw counter;
 * both are advanced by one. table [rows, 8] (nullable): the step scalars of the coming iterations as the HOST computed them
 * (Adam step size and 1 / sqrt(bias_correction2) per optimiser group — torch/optim/adam.py —, the proposal weight anneal of
 * models/nerfacto.py:270-280, the batch slot); row counter[0] % rows is copied to hyper [8]. uniform0 [n0], uniform1 [n1]
 * (nullable with n = 0): U[0, 1) draws of this step — what the reference takes from torch.rand for the samplers' jitter
 * (ray_samplers.py:105, 322) and the random background (renderers.py:195) — Philox-4x32-10 keyed by (seed, counter[1]): the same
 * numbers whether the step is launched eagerly or replayed. */
int nsamd_step_prologue(int64_t* counter, const float* table, int32_t rows, float* hyper, float* uniform0, int64_t n0,
                        float* uniform1, int64_t n1, uint64_t seed, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Camera-pose corrections (CameraOptimizer, cameras/camera_optimizers.py:85-185; exponential maps cameras/lie_groups.py:
 * 25-60 SO3xR3, :63-117 SE3). pose [num_cameras,6] = (translation, rotation vector) per camera, mode 1 = SO3xR3, 2 = SE3.
 *   nsamd_camera_apply     apply_to_raybundle (:148-153): origins = raw_origins + t(c), directions = R(c) raw_directions
 *                          for the camera c of every ray; fp32, the reference's operations in the reference's order.
 *   nsamd_camera_backward  what autograd does with dL/d(origins, directions) of those rays — the index_add over the rays
 *                          of a camera, bmm backward, the exponential map's backward — plus the gradient of the L2
 *                          regulariser (:179-185): dpose [num_cameras,6] += d(loss + regulariser)/d pose; *regulariser
 *                          (nullable) = mean|t| trans_l2_penalty + mean|w| rot_l2_penalty. `upstream` lists the per-ray
 *                          gradient buffers of the sampling levels that saw the rays (nsamd_hashgrid_encode_bwd_rays,
 *                          one pair per level, summed per ray in list order). Per-camera sums in double in a fixed order:
 *                          bit-reproducible. `non_trainable_camera_indices` is not supported here (callers keep torch).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct nsamd_ray_grads {
  const float* d_origins[4];    /* [n,3] each */
  const float* d_directions[4]; /* [n,3] each */
  int32_t count;
} nsamd_ray_grads;
int nsamd_camera_apply(const float* pose, int32_t mode, int32_t num_cameras, const float* raw_origins,
                       const float* raw_directions, const int64_t* camera_indices, int64_t n, float* origins,
                       float* directions, nsamd_stream_t stream);
int nsamd_camera_backward(const float* pose, int32_t mode, int32_t num_cameras, const float* raw_directions,
                          const int64_t* camera_indices, int64_t n, nsamd_ray_grads upstream, float trans_l2_penalty,
                          float rot_l2_penalty, float* dpose, float* regulariser, nsamd_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused Adam over a flat fp32 arena (engine/optimizers.py:74-193 with AdamOptimizerConfig(lr, eps=1e-15),
 * torch.optim.Adam semantics: bias-corrected, no weight decay, no amsgrad). grad_scale multiplies the gradient
 * first (1/world_size for the data-parallel mean, or the inverse loss scale). step is 1-based.
 * The arithmetic follows torch/optim/adam.py (_single_tensor_adam) operation by operation on fp32, scalars evaluated in double
 * and rounded once (hence the double arguments): the updated parameters and moments are the same bits as torch.optim.Adam's.
 * hyper_dev (nullable): device floats {lr / (1 - beta1^step), sqrt(1 - beta2^step)} overriding the values derived
 * from (lr, step) — lets a captured hipGraph be replayed across steps.
 * ------------------------------------------------------------------------------------------------------------ */
int nsamd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, int32_t step, float grad_scale, const float* hyper_dev,
                    nsamd_stream_t stream);

/* Library / device introspection. */
const char* nsamd_version(void);
const char* nsamd_status_string(int status);
int nsamd_device_info(int32_t* num_cus, int32_t* wavefront_size, int32_t* lds_bytes_per_cu, char* arch_name,
                      int32_t arch_name_len);

/* MFMA lane-layout probe used by the tests: out[16,16] = A[16,4] * B[4,16] through one v_mfma_f32_16x16x4_f32
 * with the operand/result lane mapping the field kernels assume. */
int nsamd_probe_mfma16(const float* A, const float* B, float* out, nsamd_stream_t stream);
/* The same for v_mfma_f32_16x16x32_bf16 (A [16,32], B [32,16] fp32 holding bf16-representable values): the lane mapping of
 * the field backward's weight-gradient GEMMs on two-piece bf16 operands. */
int nsamd_probe_mfma_bf16(const float* A, const float* B, float* out, nsamd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NSAMD_H */
