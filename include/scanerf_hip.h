/*
 * scanerf_hip.h -- C ABI of libscanerf_hip.so: the MI355X (gfx950) implementation of
 * ScaNeRF's per-tile volume-rendering hot path.
 *
 * Drop-in boundary: one entry point per op of the reference's two pybind modules
 * (CUDA_EXT: cuda/binding.cpp:10-54, HASHGRID: hashgrid/binding.cpp:9-44) that lies on
 * the hot path, plus the fused fast-path entry points.  Plain pointers and sizes only
 * (no torch types).  Every pointer is a DEVICE pointer unless marked [host].  All
 * tensors are contiguous row-major.  Outputs are pre-allocated and pre-filled by the
 * caller exactly as the reference's Python does (sentinel -1 / zeros); the library
 * never allocates and keeps no state between calls.
 *
 * Every function returns 0 on success, non-zero on failure; scanerf_last_error()
 * returns a thread-local message for the last failure.  Launches are asynchronous on
 * `stream` (a hipStream_t passed as void*; NULL = the null stream); no call synchronises.
 *
 * Citations are file:line under the reference repository.
 */
#ifndef SCANERF_HIP_H_
#define SCANERF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *scanerf_stream_t; /* hipStream_t */

#define SCANERF_PARAMSIZE 13994 /* hashgrid/include/decoder.h:20 */

/* feature-table element types (scanerf_*: `feat_dtype`) */
#define SCANERF_F32 0
#define SCANERF_F16 1
#define SCANERF_BF16 2

const char *scanerf_last_error(void);
int scanerf_abi_version(void); /* 9 */
/* Options.  The library keeps no global or per-process state and reads no environment variable: what a caller may choose is an
 * argument of the call it affects --
 *   decoder arithmetic of the fused training kernels      scanerf_render_cfg.arith (SCANERF_ARITH_*)
 *   levels masked by the coarse-to-fine schedule          scanerf_render_cfg.skip_levels
 *   decoder arithmetic of the render-time inference ops   SCANERF_INFER_H3 / SCANERF_INFER_F32 in `sample_major`
 *   per-sample array layout of the render-time ops        `sample_major` (0 / 1 / 2), SCANERF_SKIP_UNSAMPLED
 *   table-gradient record format of the stand-alone op    `compact_records` (-1 default / 0 / 1 / 2)
 *   large-table route (dfeat + stand-alone scatter, or the backward's own records + split pass): which entry points are called
 * Launch shapes, alternative producers and timing builds of the A/B experiments exist only in a library built with
 * `make EXP=1` (-DSCANERF_EXPERIMENTS), where they are read from the environment; scanerf_experiments_enabled() tells which
 * build this is (0 = product build). */
int scanerf_experiments_enabled(void);

/* ---- CUDA_EXT surface ------------------------------------------------------------ */

/* cuda/compute_ray_kernel.cu:95-115 (compute_ray.h:9-14).  locs [B,3] i32 (view,px,py),
 * Ks [C,9], C2Ws [C,12] -> rays_o, rays_d [B,3] */
int scanerf_compute_ray_forward(float *rays_o, float *rays_d, const float *Ks, const float *C2Ws,
                                const int32_t *locs, int B, scanerf_stream_t stream);

/* cuda/compute_ray_kernel.cu:117-136 (compute_ray.h:16-21).  grad_C2Ws [C,12] is ACCUMULATED
 * into.  Implements the correct adjoint (per-ray gradients); the reference reads
 * grad_rays_*[view_idx] (compute_ray_kernel.cu:71-72), see DESIGN.md. */
int scanerf_compute_ray_backward(const float *grad_rays_o, const float *grad_rays_d, const float *Ks,
                                 float *grad_C2Ws, const int32_t *locs, int B, int num_cam,
                                 scanerf_stream_t stream);

/* cuda/helper_kernel.cu:130-148 (K=1) and :177-197 (v2).  center,size [K,3]; bounds [B,K,2] */
int scanerf_ray_aabb_intersection(const float *rays_o, const float *rays_d, const float *center,
                                  const float *size, float *bounds, int B, int K, scanerf_stream_t stream);

/* cuda/helper_kernel.cu:645-671: the live training sampler.  occ = bool grid [2^lx,2^ly,2^lz]
 * (1 byte/cell), log2dim [3] i32, corner/size [3] (all device); z_vals,dists [B,S] pre-filled -1. */
int scanerf_sample_points_grid(const float *rays_o, const float *rays_d, float *z_vals, float *dists,
                               const float *block_corner, const float *block_size, const uint8_t *occ,
                               const int32_t *log2dim, int B, int S, scanerf_stream_t stream);

/* cuda/sample_kernel.cu:102-126.  Rays that miss the box are left untouched (the reference
 * device-asserts); *missed (device i32, may be NULL) counts them. */
int scanerf_sample_insideout_block(const float *rays_o, const float *rays_d, int S, int S_bg,
                                   const float *block_center, const float *block_size, float far_,
                                   float *z_vals, float *z_vals_bg, int32_t *missed, int B,
                                   scanerf_stream_t stream);

/* cuda/sample_kernel.cu:48-68 */
int scanerf_background_sampling(const float *starts, const float *bg_depth, float *z_vals, int S,
                                float sample_range, int B, scanerf_stream_t stream);

/* cuda/adam_kernel.cu:72-94 / :147-169.  Arrays are [K,8] (index = row*8+dim, dim<param_dim);
 * `step` is the PREVIOUS step count (the kernel uses step+1: adam.h:16 `int &step` quirk).
 * Entries whose grad is exactly 0 are skipped (no moment decay).  fp16: moments are IEEE half. */
int scanerf_adam_step(float *params, const float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                      float beta1, float beta2, float eps, int step, int64_t K, int param_dim,
                      scanerf_stream_t stream);
int scanerf_adam_step_fp16(float *params, const float *grad, void *exp_avg, void *exp_avg_sq, float lr,
                           float beta1, float beta2, float eps, int step, int64_t K, int param_dim,
                           scanerf_stream_t stream);

/* ---- HASHGRID surface ------------------------------------------------------------ */

/* hashgrid/src/hashgrid_bg_kernel.cu:229-249.  points [N,3] in [-2,2]; features [L,T,2]
 * (feat_dtype); resolutions [L,3] i32; outputs [N,L,2] f32.  T must be a power of two. */
int scanerf_embedding_bg_forward(const float *points, float *outputs, const void *features,
                                 const int32_t *resolutions, int N, int L, int T, int feat_dtype,
                                 scanerf_stream_t stream);
/* hashgrid/src/hashgrid_bg_kernel.cu:251-275.  grad_points [N,3] and grad_features [L,T,2]
 * (f32) are ACCUMULATED into (caller zero-fills: PyHashGridBG.py:27-28).  Either may be NULL. */
int scanerf_embedding_bg_backward(const float *points, const float *grad_in, float *grad_points,
                                  float *grad_features, const float *features, const int32_t *resolutions,
                                  int N, int L, int T, scanerf_stream_t stream);
/* Domain: as the reference's kernels (hashgrid_bg_kernel.cu:107-150 does not clamp), every encoder op takes points in [-2, 2]^3.
 * Outside, the plain ops extrapolate like the reference (negative / > 1 interpolation weights); the COMPACT table-gradient records
 * (8- and 12-byte formats: scanerf_embedding_bg_backward_binned[_adam] with compact_records 1 / 2 or the default for point-major
 * rows, the fused path behind SCANERF_ARITH_T16 / T16S, scanerf_table_grad_scatter_adam_rays) store the x-weight as an unsigned
 * fraction in [0, 1) and are defined for in-domain points only -- which is what the samplers and the two contractions produce. */
/* The same table gradient without global atomics: radix partition of the contributions by
 * 2048-entry table bucket + LDS accumulation (csrc/scatter.hip).  grad_layout 0: grad_in is
 * [N][L][2] (binding surface), 1: [L][N][2].  workspace: caller-owned scratch of at least
 * scanerf_embedding_bwd_workspace_bytes(N,L,T) bytes, 16-byte aligned (0 => shape not
 * supported by the binned path; use scanerf_embedding_bg_backward).  Does not compute
 * grad_points (call scanerf_embedding_bg_backward with grad_features = NULL for that). */
size_t scanerf_embedding_bwd_workspace_bytes(int N, int L, int T);
int scanerf_embedding_bg_backward_binned(const float *points, const float *grad_in, float *grad_features,
                                         const int32_t *resolutions, int N, int L, int T, int grad_layout,
                                         void *workspace, size_t workspace_bytes,
                                         int compact_records /* record format: -1 = the layout's default (12-byte records for point-major
                                            rows of 16 levels, 16-byte ones for level-major gradients), 0 = 16-byte (exact products),
                                            1 = 8-byte (level-major only), 2 = 12-byte: csrc/scatter_common.h */,
                                         scanerf_stream_t stream);
/* The same ending in the fused sparse Adam (the bucket images are the touched-entry list: no gradient table, no zero-fill, no
 * scan; per element the IEEE sequence of scanerf_adam_step, `step` = previous count).  half_table (may be NULL): f16 / bf16
 * gather copy refreshed where params change.  overflow_grad: zero [L][T][2] f32 table, written (and consumed) only if the
 * workspace overflows.  compact_records (grad_layout 1): 1 = 8-byte records (13-bit significands, csrc/scatter_common.h) -- for
 * feature gradients that come out of the t16 backward's f16 products; 2 = 12-byte records (f32 components less 4 bits, 23-bit
 * weight: behind the t16s backward); 0 = 16-byte records, exact. */
int scanerf_embedding_bg_backward_binned_adam(const float *points, const float *grad_in, const int32_t *resolutions, int N,
                                              int L, int T, int grad_layout, void *workspace, size_t workspace_bytes,
                                              float *params, float *exp_avg, float *exp_avg_sq, void *half_table,
                                              int half_dtype, float *overflow_grad, float lr, float beta1, float beta2,
                                              float eps, int step, int compact_records, scanerf_stream_t stream);
/* The same scatter + sparse Adam for the samples of up to TWO render branches over the same B rays (a tile's foreground and
 * background, tile.py:639-692 -- one optimiser step for both, tile.py:1010), without materialising or concatenating contracted
 * points: point positions are formed from rays_o/rays_d [B,3] and each branch's depths z [B,S] exactly as the render kernels form
 * them (contract_mode 0 = contract_fore, 1 = contract_bg; min_bbox / bbox_size [3] on the HOST), gradients are each branch's
 * level-major dfeat [16][B*S][2] out of scanerf_render_backward; valid [B] u8 (may be NULL) masks rays out.  z2 = NULL: one branch.
 * Tables of >= 2^22 entries per level, L = 16, 12-byte records in 64-byte segments (csrc/scatter_common.h format 3); workspace
 * of scanerf_embedding_bwd_workspace_bytes(B*(S1+S2), 16, T).  Replaces hashgrid_bg_kernel.cu:152-226 + torch.optim.Adam. */
int scanerf_table_grad_scatter_adam_rays(const float *rays_o, const float *rays_d, int B, const float *z1, const float *dfeat1,
                                         const uint8_t *valid1, int S1, int contract_mode1, const float *z2, const float *dfeat2,
                                         const uint8_t *valid2, int S2, int contract_mode2, const float *min_bbox /*[host]*/,
                                         const float *bbox_size /*[host]*/, const int32_t *resolutions, int T, void *workspace,
                                         size_t workspace_bytes, float *params, float *exp_avg, float *exp_avg_sq, void *half_table,
                                         int half_dtype, float *overflow_grad, float lr, float beta1, float beta2, float eps, int step,
                                         int fp16_moments /* OPT-IN, default 0: exp_avg / exp_avg_sq are [16][T][2] HALF arrays and the
                                            update is adam_step_cuda_fp16's (cuda/adam_kernel.cu:98-144: gradient x 128, moments stored in
                                            half): 16 instead of 24 bytes of optimiser state moved per touched entry each way.  NOT what the
                                            reference's live code runs (torch.optim.Adam, fp32 state, tile.py:301) */,
                                         scanerf_stream_t stream);
/* hashgrid/src/hashgrid_kernel.cu:246-270 / :272-300 (world-space box variant) */
int scanerf_embedding_forward(const float *points, float *outputs, const float *features,
                              const float *block_corner, const float *block_size,
                              const int32_t *resolutions, int N, int L, int T, scanerf_stream_t stream);
int scanerf_embedding_backward(const float *points, const float *grad_in, float *grad_points,
                               float *grad_features, const float *features, const float *block_corner,
                               const float *block_size, const int32_t *resolutions, int N, int L, int T,
                               scanerf_stream_t stream);

/* ---- fused fast path -------------------------------------------------------------- */

/* One launch for hashgrid/__init__.py:512-596 (HashGrid.render_batch_rays): for each ray,
 * samples o+z*d -> contraction (mode 0: contract_fore :394-395, 1: contract_bg :397-411) ->
 * 16-level hash lookup -> ShallowMLP (network.py:172-190) -> cal_integrate_weight/accumulate
 * (:344-366).  Nothing per-sample is materialised except `weights`.
 *
 *   rays_o, rays_d [B,3]; z_vals, dists [B,S]; features [16,T,2] (feat_dtype);
 *   resolutions [16,3] i32; mlp_blob [13994] f32 in decoder.h blob order (rendering.py:101-112);
 *   weight_feature [32] f32 (hashgrid/__init__.py:228-235 repeated x2);
 *   min_bbox, bbox_size [3] [host] (the HashGrid 2x box);
 *   out_ray [B,16] f32: rgb(3, clamped) depth(1) T_left(1) diffuse(3) specular(3) tint(3)
 *                       sum_w_spec2 (1: sum_i w_i*|c_s,i|^2, the l2_reg_specular numerator) pad(1);
 *   weights [B,S] f32 (may be NULL).
 */
typedef struct {
    int contract_mode; /* 0 fore, 1 bg */
    int infinity;      /* dists[:, -1] = 1e10 (hashgrid/__init__.py:349-350) */
    float min_bbox[3];
    float bbox_size[3];
    int arith; /* decoder arithmetic: SCANERF_ARITH_F32 (f32-input MFMA, exact f32), SCANERF_ARITH_H3 (f16 MFMA
                  on hi/lo-split operands, three products per term: f32-equivalent results, csrc/render_h3.h) or
                  SCANERF_ARITH_T16 (forward as H3; backward on 16-sample tiles at two waves per SIMD with the
                  forward recompute in H3 and the gradient products on one f16 MFMA per term, csrc/render_t16.h;
                  needs the x-stash) or SCANERF_ARITH_T16S (the T16 kernel with every gradient product split as well --
                  three MFMAs per term, G' in f32, f32 table-gradient records: f32-equivalent gradients at the T16
                  structure's speed; needs the x-stash).  plan / backward / accumulate of one step must be given the same value */
    unsigned skip_levels; /* bit l set: level l's inputs meet exactly-zero first-layer weights (the coarse-to-fine mask,
                             hashgrid/__init__.py:228-235, folded into the packed decoder), so scanerf_render_forward* may
                             leave the level's table alone (its encoder outputs become 0: same results bit for bit).  0 = none */
} scanerf_render_cfg;
#define SCANERF_ARITH_F32 0
#define SCANERF_ARITH_H3 1
#define SCANERF_ARITH_T16 2
#define SCANERF_ARITH_T16S 3

/* Packs the decoder blob (+ weight_feature folded into the first layer) into the LDS image
 * the fused kernels stage (csrc/render_common.h).  workspace: scanerf_render_workspace_floats()
 * f32, 16-byte aligned, caller-owned; re-pack whenever the blob or weight_feature changes. */
int scanerf_render_workspace_floats(void);
int scanerf_pack_decoder(const float *mlp_blob, const float *weight_feature, float *workspace,
                         scanerf_stream_t stream);

/* ray_valid [B] u8 (may be NULL): rays with 0 render as zeros with T_left = 1
 * (hashgrid/__init__.py:427-431) and are skipped.  tile_T [B, ceil(S/16)] (may be NULL): the
 * transmittance entering each 16-sample tile, saved for scanerf_render_backward.  xstash [B*S,32] f32
 * (may be NULL): the encoder outputs in register order, so the backward can skip the re-gather. */
int scanerf_render_forward_packed(const float *rays_o, const float *rays_d, const float *z_vals,
                                  const float *dists, const void *features, int feat_dtype,
                                  const int32_t *resolutions, const float *workspace,
                                  const scanerf_render_cfg *cfg /*[host]*/, const uint8_t *ray_valid,
                                  float *out_ray, float *weights, float *tile_T, float *xstash, int B, int S,
                                  int T, scanerf_stream_t stream);
/* scanerf_render_forward_packed that ALSO does scanerf_render_scatter_plan's work for the t16 backward of the same rays: the
 * record counts come out of the forward kernel (its hash indices are the plan's), then the scans run.  Call it INSTEAD of
 * scanerf_render_scatter_plan, with that function's workspace and cfg->arith = SCANERF_ARITH_T16; only for shapes where
 * scanerf_render_forward_plan_supported() is non-zero (forward and backward kernels visit the same rays per workgroup). */
int scanerf_render_forward_packed_plan(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                                       const void *features, int feat_dtype, const int32_t *resolutions, const float *workspace,
                                       const scanerf_render_cfg *cfg, const uint8_t *ray_valid, float *out_ray, float *weights,
                                       float *tile_T, float *xstash, void *jstash /* [B,ceil(S/32),8,4,64] u32: six 20-bit significands + one exponent per (sample, level), or NULL (fp32 tables) */,
                                       int B, int S, int T, void *scatter_ws, size_t scatter_ws_bytes, scanerf_stream_t stream);
int scanerf_render_forward_plan_supported(int B, int S, int T);

/* Adjoint of scanerf_render_forward_packed (hashgrid/__init__.py:512-596 under autograd).
 *   out_ray, tile_T: the forward's outputs;  grad_out [B,16]: dL/d(out_ray) (columns as out_ray;
 *   the sum_w_spec2 column is differentiated with the weights detached, as the reference's
 *   l2_reg_specular, hashgrid/__init__.py:593);
 *   dfeat [16][B*S][2] f32 (out): dL/d(hash features), level-major -> feed to
 *   scanerf_embedding_bg_backward_binned(grad_layout = 1) with the contracted sample points; may be
 *   NULL when scatter_ws is given;
 *   scatter_ws / scatter_ws_bytes / grad_features (may be NULL/0/NULL): FUSED table-gradient path.  The
 *   kernel then appends the scatter records itself (no dfeat round trip, no separate producer pass):
 *     scanerf_render_scatter_plan(...)        -- before: reserves each workgroup's record ranges
 *     scanerf_render_backward(..., ws, ...)   -- emits
 *     scanerf_render_scatter_accumulate(...)  -- after: grad_features [16][T][2] += records
 *   all three on the same (B, S, T), rays, z_vals, cfg, ray_valid and stream; grad_features is written
 *   by the backward only if the workspace overflows;
 *   dw_partial [4*scanerf_render_backward_grid(B)][13994] f32 scratch;
 *   grad_blob [13994] f32: dL/d(decoder blob), ACCUMULATED into (deterministic reduction). */
int scanerf_render_backward_grid(int B);
int scanerf_render_backward(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                            const void *features, int feat_dtype, const int32_t *resolutions,
                            const float *workspace, const float *weight_feature,
                            const scanerf_render_cfg *cfg /*[host]*/, const uint8_t *ray_valid,
                            const float *out_ray, const float *tile_T, const float *grad_out,
                            const float *xstash /* forward's, or NULL */, float *dfeat, float *dw_partial,
                            float *grad_blob, float *g_dnorm /* [B,ceil(S/32)] or NULL */,
                            float *g_rowsum /* [B,2,64] or NULL */,
                            const void *jstash /* forward's position Jacobians [B,ceil(S/32),8,4,64] u32, or NULL */,
                            float *g_raypos /* [B,6] dL/d(rays_o), dL/d(rays_d) through the sample positions (t16, needs jstash,
                                               g_dnorm and g_rowsum; zero-filled by the caller), or NULL */,
                            void *scatter_ws, size_t scatter_ws_bytes,
                            float *grad_features, int B, int S, int T, scanerf_stream_t stream);
/* Fused table-gradient path (replaces the atomicAdd scatter of hashgrid_bg_kernel.cu:196-201 for the
 * samples of a render batch).  workspace_bytes == 0 => shape unsupported, use dfeat + the binned op.
 * Precision of the sum: records are added in 64-bit fixed point (bit-reproducible).  Tables of at most 2^21 entries per level:
 * the records carry what the emitting backward wrote (f32 behind SCANERF_ARITH_F32 / H3, 19-bit-mantissa f32 behind T16S,
 * 13-bit behind T16).  Tables ABOVE 2^21 entries, when the workspace has the size scanerf_render_scatter_workspace_bytes
 * returns (it then holds the split pass's second record area): every record -- also an f32 one out of the F32 / H3 backward
 * -- is re-encoded by the split pass as a 12-byte record (gradient components rounded to 19 mantissa bits, x-weight to 23
 * bits: ~2^-21 relative per record against the window path), and the second entry of an x-neighbour pair that crosses a
 * 2^13-entry window (levels with a resolution above 8 192 only) is added to grad_features / overflow_grad with float atomics,
 * i.e. NOT bit-reproducibly; a smaller caller-owned workspace (or SCANERF_NO_SPLIT=1 in an experiments build) keeps the window
 * re-reads (f32 records, reproducible, slower). */
size_t scanerf_render_scatter_workspace_bytes(int B, int S, int T);
int scanerf_render_scatter_plan(const float *rays_o, const float *rays_d, const float *z_vals,
                                const int32_t *resolutions, const scanerf_render_cfg *cfg /*[host]*/,
                                const uint8_t *ray_valid, int B, int S, int T, void *workspace,
                                size_t workspace_bytes, scanerf_stream_t stream);
int scanerf_render_scatter_accumulate(float *grad_features, int B, int S, int T, void *workspace,
                                      size_t workspace_bytes, scanerf_stream_t stream);
/* accumulate + fused sparse Adam in one pass (replaces scanerf_render_scatter_accumulate + scanerf_adam_step on the table:
 * the bucket images ARE the touched-entry list, so no gradient table, no zero-fill, no scan; same per-element IEEE sequence as
 * scanerf_adam_step / cuda/adam_kernel.cu:24-69, `step` = the previous step count).  params / exp_avg / exp_avg_sq [16][T][2]
 * f32; half_table (may be NULL): f16 / bf16 (half_dtype = SCANERF_F16 / SCANERF_BF16) gather copy of params, refreshed where
 * params change; overflow_grad (may be NULL): the table given to scanerf_render_backward as grad_features -- consulted and
 * re-zeroed only if the record workspace overflowed, so it is allocated zero once and never filled per step. */
int scanerf_render_scatter_accumulate_adam(float *params, float *exp_avg, float *exp_avg_sq, void *half_table,
                                           int half_dtype, float *overflow_grad, float lr, float beta1, float beta2,
                                           float eps, int step, int B, int S, int T, void *workspace,
                                           size_t workspace_bytes, scanerf_stream_t stream);
/* ... over TWO record sets (a tile's foreground and background branch, each planned / emitted on its own workspace over the
 * same B rays and table, S1 / S2 samples): both gradients meet in one bucket image and ONE Adam step (tile.py:880-1015). */
int scanerf_render_scatter_accumulate_adam2(float *params, float *exp_avg, float *exp_avg_sq, void *half_table,
                                            int half_dtype, float *overflow_grad, float lr, float beta1, float beta2,
                                            float eps, int step, int B, int T, int S1, void *workspace1,
                                            size_t workspace1_bytes, int S2, void *workspace2, size_t workspace2_bytes,
                                            scanerf_stream_t stream);
/* Valid-ray compaction of a training batch (hashgrid/__init__.py:419-434: valid = all(z_vals != -1), then rays_o[valid],
 * rays_d[valid], z_vals[valid], dists[valid]; tile.py gathers the targets the same way), without torch's nonzero + gathers:
 * scanerf_ray_valid writes the flags, scanerf_compact_rays moves the valid rays' entries, in their original order, to rows
 * [0, *count) of the out_* buffers (each sized for B rays; count: device int32).  target / out_t and out_index [B] i32 (source
 * ray of each output row) may be NULL.  valid: 16-byte aligned.  Wave ballot + popcount prefix sums, no atomics. */
int scanerf_ray_valid(const float *z_vals, uint8_t *valid, int B, int S, scanerf_stream_t stream);
int scanerf_compact_rays(const uint8_t *valid, int B, int S, const float *rays_o, const float *rays_d, const float *target,
                         const float *z_vals, const float *dists, float *out_o, float *out_d, float *out_t, float *out_z,
                         float *out_dist, int32_t *out_index, int32_t *count, scanerf_stream_t stream);

/* Photometric loss of the training step and dL/d(out_ray) in two launches (criterions.py:90,142-144 MSE over the valid
 * rays' rgb + tile.py:999 reg_weight * l2_reg_specular = mean over valid rays x 3 of out_ray[:,14]):
 *   loss [1] = (sum_valid |rgb - target|^2 + reg_weight * sum_valid out_ray[:,14]) / (3 * n_valid)
 *   grad_out [B,16] = its gradient w.r.t. out_ray (all 16 columns written; invalid rays zero)
 * scratch: scanerf_photometric_loss_scratch_floats() f32.  Deterministic (fixed reduction order). */
/* Ray gradients of a scanerf_render_backward call that produced g_raypos (render.py ray_gradients_fused; replaces autograd through
 * network.py:38-77 sh_encoding and the direction normalisation of :177): g_o = g_raypos[:,0:3]; g_d = g_raypos[:,3:6] + the |d|
 * path (g_dnorm [B,ceil(S/32)] summed) + the SH path (g_rowsum [B,2,64] x Directional_MLP.mlp.0 weight rows 32..47 of mlp_blob).
 * Rays with ray_valid == 0 get zeros. */
int scanerf_ray_grad_epilogue(const float *rays_d, const float *mlp_blob, const float *g_raypos, const float *g_dnorm,
                              const float *g_rowsum, const uint8_t *ray_valid, float *g_o, float *g_d, int B, int S,
                              scanerf_stream_t stream);
/* Alpha compositing of per-sample decoder outputs along rays and its adjoint, as stand-alone ops: HashGrid.cal_integrate_weight
 * (hashgrid/__init__.py:344-360) + the accumulate calls and the l2_reg_specular sum of render_batch_rays (:362-366, :564-574,
 * :591-594) in one launch each way, for callers that keep the decoder outputs as tensors (the op-by-op route).
 *   sigma [B*S], diffuse / specular / tint [B*S,3], z_vals / dists [B,S], rays_d [B,3]; infinity: the last sample's delta is 1e10;
 *   out_ray [B,16]: the columns scanerf_render_forward_packed writes (rgb, depth, T_left = transmittance BEFORE the last sample,
 *   diffuse, specular, tint, sum w |c_s|^2); weights [B,S] (may be NULL).
 * Backward: grad_out [B,16] (column 14 differentiates sum w |c_s|^2 with w detached, as :593), grad_weights [B,S] (may be NULL) ->
 *   dL/d(sigma) [B*S], dL/d(diffuse / specular / tint) [B*S,3], g_dnorm [B] (may be NULL) = dL/d|rays_d| through delta. */
int scanerf_composite_forward(const float *sigma, const float *diffuse, const float *specular, const float *tint, const float *z_vals,
                              const float *dists, const float *rays_d, float *out_ray, float *weights, int B, int S, int infinity,
                              scanerf_stream_t stream);
int scanerf_composite_backward(const float *sigma, const float *diffuse, const float *specular, const float *tint, const float *z_vals,
                               const float *dists, const float *rays_d, const float *out_ray, const float *grad_out,
                               const float *grad_weights, float *g_sigma, float *g_diffuse, float *g_specular, float *g_tint,
                               float *g_dnorm, int B, int S /* <= 512 */, int infinity, scanerf_stream_t stream);
int scanerf_photometric_loss_scratch_floats(void);
int scanerf_photometric_loss_grad(const float *out_ray, const float *target /*[B,3]*/, const uint8_t *ray_valid,
                                  float reg_weight, float *grad_out, float *loss, float *scratch, int B,
                                  scanerf_stream_t stream);
/* The same for the complete per-tile render (tile.py:666-690 merge, tile.py:880-1015 loss): pred = fg.rgb + fg.T_left * bg.rgb,
 * loss [1] = mean over the rays valid in either branch (criterions.py:121-138: valid = fore_valid | bg_valid) x 3 of
 * (pred - target)^2 + reg_weight * (l2_reg_specular of the fg-valid rays + of the bg-valid rays); rays invalid in both
 * branches get zero gradients; grad_fg / grad_bg [B,16] = its gradients w.r.t. the two branches' out_ray (fg: rgb, T_left and column 14;
 * bg: rgb and column 14).  scratch as above. */
int scanerf_photometric_loss_grad_fgbg(const float *out_fg, const float *out_bg, const float *target,
                                       const uint8_t *valid_fg, const uint8_t *valid_bg, float reg_weight, float *grad_fg,
                                       float *grad_bg, float *loss, float *scratch, int B, scanerf_stream_t stream);
/* Device self-test of the split-f16 backward primitives (csrc/render_h3.h; test infrastructure, one wave):
 * workspace from scanerf_pack_decoder; dy, x [64][32] f32 -> out_dx [2][64][32] (W^T dy of Spatial_MLP.mlp.2, and
 * of the H part of Directional_MLP.mlp.0 in rows 0..31 of the second slab), out_dw [64][64] = dy x^T, out_rs [64]
 * = row sums of dy. */
int scanerf_h3_selftest(const float *workspace, const float *dy, const float *x, float *out_dx, float *out_dw,
                        float *out_rs, scanerf_stream_t stream);
/* Test infrastructure: one launch of ~300 KB of straight-line code on every CU, after which no other kernel's instructions are
 * left in the instruction caches (no reference counterpart; tests/test_gpu_determinism.py, tools/fault_probe.py: faults that only
 * show on cold instruction caches). */
int scanerf_icache_sweep(scanerf_stream_t stream);
/* Measurement infrastructure (no reference counterpart): blocks x 512 threads each issue loads_per_thread (multiple of 8) scattered
 * 8-byte loads over table[0 .. entries) (entries: power of two; make it far larger than the 32 MB of L2).  Timed by the caller,
 * loads / second = the chip's L2 <-> fabric request rate for scattered accesses (bench.py `roofline.fabric_requests`). */
int scanerf_gather_rate_probe(const void *table, long long entries, int blocks, int loads_per_thread, unsigned *sink,
                              scanerf_stream_t stream);
/* Test infrastructure: the 8-byte scatter-record codec (csrc/scatter_common.h Rec8) on n values.  words [n][2] = the packed
 * records; out [n][8] = l0, l1, the four contributions the accumulate adds (x, y to l0; x, y to l1), E - 25, t. */
int scanerf_rec8_selftest(const float *gx, const float *gy, const float *tx, const uint32_t *l0, const uint32_t *k, int n,
                          uint32_t *words, float *out, scanerf_stream_t stream);
/* For pose refinement (gradients w.r.t. the rays): g_dnorm = per-tile partials of dL/d|rays_d| through
 * delta = dist*|d| (hashgrid/__init__.py:347); g_rowsum = per-ray sums of dL/d(Directional_MLP.mlp.0
 * pre-activation) in two partial rows (their sum times W[:,32:48] is dL/dSH(viewdir)).  The gradient
 * through the sample positions comes from scanerf_embedding_bg_point_grad on dfeat. */
int scanerf_embedding_bg_point_grad(const float *points, const float *dfeat_level_major, float *grad_points,
                                    const float *features, const int32_t *resolutions, int N, int L, int T,
                                    scanerf_stream_t stream);

/* Encoder with an explicit mapping (for benchmarks and the two-kernel path):
 * variant 0 auto / 1 XCD-partitioned by level / 2 level-fastest; level_major_out != 0 writes
 * [L][N][2] instead of the binding surface's [N][L][2]. */
int scanerf_embedding_bg_forward_ex(const float *points, float *outputs, const void *features,
                                    const int32_t *resolutions, int N, int L, int T, int feat_dtype,
                                    int variant, int level_major_out, scanerf_stream_t stream);

/* ---- HASHGRID surface, render-time ops (hashgrid/include/rendering.h:20-182;
 *      hashgrid/src/rendering_kernel.cu, file:line per function) --------------------------------
 * Tile set: corners/sizes [nb,3] f32; occ = concatenated bool grids; grid_starts [nb] i64;
 * log2dim [nb,3] i32; intersections [B,nb,2] f32 (1e7 = miss); tracing_blocks [B,nb] i32
 * (argsort of near); block indices are int16, 4 slots, -1 = none. */
int scanerf_ray_block_intersection(const float *rays_o, const float *rays_d, const float *corners,
                                   const float *sizes, float *intersections, int B, int nb,
                                   scanerf_stream_t stream);                                   /* :126-174 */
/* sample_major (the per-sample arrays of the render-time ops below): 0 = the reference's [B][S] arrays; 1 = [S][B] (z_vals,
 * dists, block_idxs [S][B][4], the per-sample outputs [S][B][3] / [S][B]): the renderer's own layout -- a wave's 32 samples are
 * then one depth index of 32 neighbouring rays (pixels), whose cells coincide down to the fine levels, instead of 32 depths of
 * one ray.  Same arithmetic per sample. */
int scanerf_render_sample_points(const float *rays_o, const float *rays_d, const float *corners,
                                 const float *sizes, const uint8_t *occ, const int64_t *grid_starts,
                                 const int32_t *log2dim, const int32_t *tracing_blocks,
                                 const float *intersections, int32_t *tracing_idx, float *z_start,
                                 float *z_vals, float *dists, int B, int S, int nb,
                                 int sample_major, scanerf_stream_t stream);                                     /* :179-382 */
int scanerf_prepare_points(const float *z_vals, const uint8_t *running_mask, const float *intersections,
                           int16_t *block_idxs, int B, int S, int nb, int sample_major, scanerf_stream_t stream); /* :391-449 */
/* images [nb][scanerf_render_workspace_floats()]: each tile's decoder blob packed by
 * scanerf_pack_decoder with weight_feature == 1; tables [nb,16,T,2] f16; resolution [nb,16,3] i32 */
int scanerf_pts_inference(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                          const int16_t *block_idxs, const void *tables_f16, const float *images,
                          const int32_t *resolution, const uint8_t *occ, const int64_t *grid_starts,
                          const int32_t *log2dim, const float *corners, const float *sizes, float *diffuse,
                          float *specular, float *alpha, int B, int S, int T, int nb,
                          int sample_major, scanerf_stream_t stream);                                            /* :467-621 */
/* SCANERF_SKIP_UNSAMPLED, OR-ed into `sample_major` of scanerf_pts_inference_tracing and scanerf_accumulate_color (the renderer's own
 * pair; no reference counterpart): a ray whose FIRST depth is -1 holds no sample in this tracing pass (sample_points fills a ray's
 * depths from index 0) -- the inference leaves its outputs unwritten and the accumulation does not read them, instead of 28 bytes
 * of zeros written and read per sample slot of such a ray.  Same per-ray results. */
#define SCANERF_SKIP_UNSAMPLED 4
/* Decoder arithmetic of scanerf_pts_inference / scanerf_pts_inference_tracing / scanerf_bg_pts_inference_v2, OR-ed into their
 * `sample_major` as well: none = 16-sample tiles at four waves per SIMD, every product on split-f16 MFMA (default; 1e-4 of the
 * reference); SCANERF_INFER_H3 = the 32-sample-tile kernel at two waves per SIMD (same arithmetic, round 4's kernel);
 * SCANERF_INFER_F32 = the single-pass f32-input MFMA kernel (exact f32; [B][S] arrays only, also what > 64 tiles fall back to). */
#define SCANERF_INFER_H3 8
#define SCANERF_INFER_F32 16
/* SCANERF_INFER_FOLDED (16-sample-tile kernel only): `images` were packed by scanerf_pack_decoder from blobs whose three
 * Gaussian-activated layers -- Spatial_MLP.mlp.0, Directional_MLP.mlp.0 and .2, biases and weights: blob floats [0, 2112),
 * [6503, 9639), [9639, 13799) -- were multiplied by sqrt(50 log2 e) = 8.4932184 beforehand; the kernel then evaluates the
 * activation exp(-u^2 / 0.02) as exp2(-(u')^2): one vector instruction less per activation (48 per lane and 16-sample tile). */
#define SCANERF_INFER_FOLDED 32
/* prepare_points + pts_inference as ONE launch (no reference counterpart; the renderer's own route): the slot lists are derived
 * in the kernel from running_mask [B] and intersections [B,nb,2] at every use instead of being written and read back (8 bytes per
 * sample, once per tile step).  Same values as the two ops in sequence.  Needs the 16-sample-tile kernel (the default) and
 * nb <= 8; otherwise it fails and the caller runs the two ops. */
int scanerf_pts_inference_tracing(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                                  const uint8_t *running_mask, const float *intersections, const void *tables_f16,
                                  const float *images, const int32_t *resolution, const uint8_t *occ,
                                  const int64_t *grid_starts, const int32_t *log2dim, const float *corners,
                                  const float *sizes, float *diffuse, float *specular, float *alpha, int B, int S, int T,
                                  int nb, int sample_major, scanerf_stream_t stream);
int scanerf_accumulate_color(const float *pts_diffuse, const float *pts_specular, const float *pts_alpha,
                             float *transparency, const float *z_vals, float *diffuse, float *specular,
                             float *depth, int B, int S, int sample_major, scanerf_stream_t stream);             /* :624-702 */
int scanerf_render_inverse_z_sampling(const float *intersections, const int16_t *related_bidx, float *z_vals,
                                      float sample_range, int B, int S, int nb,
                                      int sample_major, scanerf_stream_t stream);                                /* :816-868 */
int scanerf_bg_pts_inference_v2(const float *rays_o, const float *rays_d, const float *z_vals,
                                const int16_t *bg_idxs, int step, const float *corners, const float *sizes,
                                const int32_t *resolution, const void *tables_f16, const float *images,
                                float *diffuse, float *specular, float *alpha, int B, int S, int T, int nb,
                                int sample_major, scanerf_stream_t stream);                                      /* :1012-1171 */
int scanerf_update_outgoing_bidx(const float *rays_o, const float *rays_d, const float *corners,
                                 const float *sizes, const int32_t *tracing_blocks, const float *intersections,
                                 int16_t *outgoing_bidxs, float *blend_weights, float ratio, int skip, int B,
                                 int nb, scanerf_stream_t stream);                             /* :1263-1401 */
int scanerf_update_outgoing_bidx_v2(const float *rays_o, const float *corners, const float *sizes,
                                    int16_t *inside_bidxs, float *blend_weights, int B, int nb,
                                    scanerf_stream_t stream);                                  /* :1406-1474 */
/* The tile order rendering.py:301 gets from torch.argsort(intersections[..., 0], dim=-1) (stable): order [B,nb] i32 = the
 * ray's tiles by entry distance (misses, 1e7, last; equal distances by tile index).  nb <= 64. */
int scanerf_sort_tracing_blocks(const float *inter, int32_t *order, int B, int nb, scanerf_stream_t stream);
int scanerf_get_last_block(const int32_t *tracing_blocks, int32_t *bidxs, const float *intersections, int B,
                           int nb, scanerf_stream_t stream);                                   /* :1212-1260 */
int scanerf_ray_firsthit_block(const float *rays_o, const float *rays_d, const float *corners, const float *sizes,
                               const uint8_t *occ, const int64_t *grid_starts, const int32_t *log2dim,
                               const int32_t *tracing_blocks, const float *intersections, int16_t *hit_blockIdxs,
                               int B, int nb, scanerf_stream_t stream);                        /* :705-813 */
int scanerf_process_occupied_grid(int bidx, int total_grid, const float *corners, const float *sizes,
                                  const uint8_t *occ, const int64_t *grid_starts, const int32_t *log2dim,
                                  uint8_t *tgt_occ, int nb, scanerf_stream_t stream);          /* :1479-1564 */

/* cuda/include/voxelize.h:12-119 (CUDA_EXT.voxelize_mesh after its PLY read: cuda/include/plyIO.h): mark the cells of the
 * sampler grid that the 1.5x-inflated boxes of the mesh faces overlap; init_out != 0 also marks (vis and outside) every
 * cell whose centre lies outside the union box of the faces that touch the grid.  vertices [V,3] f32, faces [F,3] i32,
 * vis / outside bool grids [2^lx,2^ly,2^lz], scratch6 6 x u32 -- device; log2dim [3], block_corner [3], block_size [3] --
 * HOST (the reference takes them as CPU tensors). */
int scanerf_voxelize_mesh(const float *vertices, const int32_t *faces, int V, int F, const int32_t *log2dim,
                          const float *block_corner, const float *block_size, uint8_t *vis, int init_out,
                          uint8_t *outside, uint32_t *scratch6, scanerf_stream_t stream);

/* ---- stand-alone decoder op (ABI 7): network.ShallowMLP.forward (network.py:172-190) and its adjoint, one launch each, on the
 * matrix cores at the fused kernels' arithmetic (split-f16 operands, three products per term, f32 accumulate: f32-equivalent).
 * What an unchanged HashGrid.render_batch_rays calls between the encoder op and its torch compositing
 * (hashgrid/__init__.py:545-548: decoder(cat([features, rays_d]), weight_feature=...)).
 *   feats / dirs: rows of 32 / 3 floats with row strides ld_feats / ld_dirs IN FLOATS -- for the reference's concatenated
 *     x [N,35]: feats = x, dirs = x + 32, both strides 35 (no slicing copy); the direction is normalised inside
 *     (d / (|d| + 1e-8), network.py:177);
 *   workspace: scanerf_pack_decoder(blob, weight_feature, workspace) -- weight_feature [32] is folded into the first layer;
 *   forward outputs: sigma [N] (= [N,1]), diffuse / specular / tint [N,3], contiguous; specular is the RAW sigmoid output
 *     (the caller multiplies by tint, hashgrid/__init__.py:568);
 *   backward: g_* = dL/d(those outputs) (contiguous; any may be NULL = zero) -> d_feats rows of 32 floats (stride ld_dfeats;
 *     dL/d(features), weight_feature applied), d_dirs rows of 3 floats (stride ld_ddirs; NULL = not wanted) -- for the
 *     gradient of x [N,35]: d_feats = gx, d_dirs = gx + 32, strides 35 -- and grad_blob [13994] += dL/d(blob) (deterministic
 *     reduction of one partial row per workgroup); dw_partial: scratch of scanerf_decoder_backward_grid(N) x 13994 floats. */
int scanerf_decoder_forward(const float *feats, int ld_feats, const float *dirs, int ld_dirs, const float *workspace,
                            float *sigma, float *diffuse, float *specular, float *tint, long long N, scanerf_stream_t stream);
int scanerf_decoder_backward_grid(long long N);
int scanerf_decoder_backward(const float *feats, int ld_feats, const float *dirs, int ld_dirs, const float *workspace,
                             const float *weight_feature, const float *g_sigma, const float *g_diffuse, const float *g_specular,
                             const float *g_tint, float *d_feats, int ld_dfeats, float *d_dirs, int ld_ddirs, float *dw_partial,
                             float *grad_blob, long long N, scanerf_stream_t stream);

#define SCANERF_RAY_OUT 16

#ifdef __cplusplus
}
#endif
#endif /* SCANERF_HIP_H_ */
