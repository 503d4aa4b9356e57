/* vidar_hip.h — C ABI of libvidar_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for ViDAR's BEV-encode -> latent-render -> chamfer hot path.
 * Every entry point takes raw DEVICE pointers + plain sizes + a hipStream_t
 * (passed as void*; NULL = the legacy default stream) and returns 0 on success,
 * a positive hipError_t code if a HIP call failed, or VIDAR_ERR_BAD_ARG.
 * No torch types cross this boundary.  All calls are asynchronous on `stream`;
 * unlike the reference (kernels on the legacy default stream followed by
 * cudaDeviceSynchronize(), e.g. third_lib/dvxlr/dvxlr.cu:510) nothing here
 * synchronises the device.  Outputs are written completely by the call
 * (including the -1 / 0 padding the reference obtains from torch::ones/zeros),
 * so callers may pass uninitialised buffers.
 *
 * Citations are file:line under the reference tree (OpenDriveLab/ViDAR @ v2).
 */
#ifndef VIDAR_HIP_H_
#define VIDAR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIDAR_ERR_BAD_ARG (-22)

/* ABI revision of this header; bumped whenever a signature changes. */
int vidar_abi_version(void);
/* profiling aid: launches the empty kernel `vidar_marker_kernel` on `stream` (see tools/trace_stats.py) */
int vidar_marker(int id, void* stream);

/* ---------------------------------------------------------------------------
 * third_lib/dvr  (pybind surface: third_lib/dvr/dvr.cpp:36-69)
 *   sigma   [N,T,Z,Y,X] f32     origin [N,TO,3] f32 (voxel units)
 *   points  [N,M,3]     f32     tindex [N,M]    f32 (frame id, <0 = padded ray)
 * ------------------------------------------------------------------------- */
int vidar_dvr_max_d(void);   /* 1446, third_lib/dvr/dvr.cu:9 */

/* dvr.render_forward(sigma, origin, points, tindex, grid, phase_name) -> [pred_dist, gt_dist]
 * third_lib/dvr/dvr.cu:65-383.  train_phase: 0 = "test", 1 = "train" (dvr.cu:361-366). */
int vidar_dvr_render_forward_f32(const float* sigma, const float* origin, const float* points,
                                 const float* tindex, float* pred_dist /*[N,M]*/,
                                 float* gt_dist /*[N,M]*/, int N, int M, int T, int TO, int Z, int Y,
                                 int X, int train_phase, void* stream);

/* dvr.render(sigma, origin, points, tindex, loss_name) -> [pred_dist, gt_dist, grad_sigma]
 * third_lib/dvr/dvr.cu:385-694.  loss_type: 0 = "l1" (and "bce"), 1 = "l2", 2 = "absrel"
 * (dvr.cu:658-672).  grad_sigma [N,T,Z,Y,X] is zeroed by the call, then accumulated with
 * fp32 hardware atomics (the reference's "+=" at dvr.cu:622 is racy). */
int vidar_dvr_render_f32(const float* sigma, const float* origin, const float* points,
                         const float* tindex, float* pred_dist, float* gt_dist, float* grad_sigma,
                         int N, int M, int T, int TO, int Z, int Y, int X, int loss_type,
                         void* stream);

/* dvr.init / dvxlr.init(points, tindex, grid) -> occupancy [N,T,Z,Y,X]
 * third_lib/dvr/dvr.cu:14-63,:705-738 == third_lib/dvxlr/dvxlr.cu:12-61,:528-561. */
int vidar_dvr_init_f32(const float* points, const float* tindex, float* occupancy, int N, int M,
                       int T, int Z, int Y, int X, void* stream);

/* ---------------------------------------------------------------------------
 * third_lib/dvxlr  (pybind surface: dvxlr.cpp:34-65, dvxlr_v2.cpp:34-70)
 * ------------------------------------------------------------------------- */
int vidar_dvxlr_max_d(void); /* 1026, third_lib/dvxlr/dvxlr.cu:10 */

/* Precision contract of the dvr / dvxlr family: the voxel traversal (tMax / tDelta, axis choice, voxel
 * indices, gt_dist) is IEEE fp64 in the reference's operation order, no FMA contraction -> index lists
 * and gt_dist are bit-exact.  The transmittance of the online ray integral is evaluated as
 * expf((float)(-cumulative_sigma_dt)) -- fp32, where the reference calls the fp64 exp -- because every
 * output it feeds (pred_dist, dd_dsigma, grad_sigma) is stored as fp32 anyway (the reference allocates
 * float outputs, dvxlr.cu:490-493); those values agree with the reference to 2e-5 relative
 * (tests/test_dvr_gpu.py), not bit for bit.  Only _f32 entry points exist for the same reason: fp64
 * tensor arguments make the reference itself fail in packed_accessor32. */

/* tuning/A-B switch of the dvr / dvxlr march launches: a launch with more than `min_waves` waves of
 * 64 rays (default 1024 = one per SIMD) ranks the rays of each 256-ray workgroup by estimated
 * length before walking them; 0 = always, INT_MAX = never.  Results do not depend on it.
 * Returns the previous value. */
int vidar_dvr_set_sort_min_waves(int min_waves);
/* which traversal the dvr / dvxlr march launches use: -1 (default) = step-parallel kernels (independent per-axis
 * tMax chains -- dvr.cu:232-251, dvxlr.cu:334-353 advance tMaxX only on X steps -- merged by exact comparisons,
 * lane-per-step integration; csrc/dvr_par.h) where they are the faster form (dvr.render at every size, the one-launch
 * dvxlr.render / render_v2 up to 98 304 rays, dvr.render_forward up to 24 576 rays), lane-per-ray kernels otherwise;
 * 0 = always lane-per-ray, 1 = always step-parallel.  Results do not depend on it: index lists and gt_dist are
 * bit-identical (tests/test_dvr_gpu.py runs every case under both).  Returns the previous value. */
int vidar_dvr_set_traversal(int mode);
/* tuning/A-B switch of dvxlr.render / render_v2: 0 = the finish pass pads the [1026] rows itself,
 * 1 (default) = one device fill ahead of the march, the finish pass only touches the live prefixes.
 * Results do not depend on it (tests/test_dvr_gpu.py runs every case under both).  Both switches are plain
 * process-wide ints read at launch time: set them before concurrent use, not during.
 * Returns the previous value. */
int vidar_dvxlr_set_pad_mode(int mode);

/* dvxlr.render(sigma, origin, points, tindex) -> [pred_dist, gt_dist, dd_dsigma, indices]
 * third_lib/dvxlr/dvxlr.cu:160-517.
 *   dd_dsigma [N,M,1026] f32 (0-padded), indices [N,M,1026,3] f32 as (z,y,x) (0-padded).
 *   The call writes every byte of the outputs (padding included); the padded row buffers must be
 *   8-byte aligned (VIDAR_ERR_BAD_ARG otherwise). */
int vidar_dvxlr_render_f32(const float* sigma, const float* origin, const float* points,
                           const float* tindex, float* pred_dist, float* gt_dist, float* dd_dsigma,
                           float* indices, int N, int M, int T, int TO, int Z, int Y, int X,
                           void* stream);

/* dvxlr.get_grad_sigma(elementwise_mult, indices, tindex, sigma_like) -> [grad_sigma]
 * third_lib/dvxlr/dvxlr.cu:63-156.  L = elementwise_mult.size(2). grad_sigma zeroed by the call.
 * `workspace` (vidar_dvxlr_get_grad_sigma_workspace_bytes(.., volumes): 1 for this call, 2 for get_grad_sigma_v2): caller-owned device
 * scratch for 8 private copies of the gradient volume -- the rays of a frame share their first voxels, whose
 * atomics otherwise serialise (one frame of 30 000 rays took as long as five); NULL = add straight into grad_sigma. */
size_t vidar_dvxlr_get_grad_sigma_workspace_bytes(int N, int T, int Z, int Y, int X, int volumes /* 1: get_grad_sigma, 2: _v2 */);
int vidar_dvxlr_get_grad_sigma_f32(const float* elementwise_mult /*[N,M,L]*/,
                                   const float* indices /*[N,M,L,3]*/, const float* tindex,
                                   float* grad_sigma /*[N,T,Z,Y,X]*/, int N, int M, int L, int T,
                                   int Z, int Y, int X, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma_regul)
 *   -> [pred_dist, gt_dist, dd_dsigma, indices, ray_pred, indicator]
 * third_lib/dvxlr/dvxlr_v2.cu:119-493. indicator is -1-padded, 1 at the first sample whose
 * exit distance reaches the un-clamped ray length (dvxlr_v2.cu:408-424). */
int vidar_dvxlr2_render_f32(const float* sigma, const float* origin, const float* points,
                            const float* tindex, const float* sigma_regul, float* pred_dist,
                            float* gt_dist, float* dd_dsigma, float* indices, float* ray_pred,
                            float* indicator, int N, int M, int T, int TO, int Z, int Y, int X,
                            void* stream);

/* dvxlr_v2.get_grad_sigma_v2(em, indices, tindex, sigma_like, indicator, grad_ray_pred)
 *   -> [grad_sigma, grad_sigma_regul]       third_lib/dvxlr/dvxlr_v2.cu:12-115 */
int vidar_dvxlr2_get_grad_sigma_f32(const float* elementwise_mult, const float* indices,
                                    const float* tindex, const float* indicator,
                                    const float* grad_ray_pred, float* grad_sigma,
                                    float* grad_sigma_regul, int N, int M, int L, int T, int Z, int Y,
                                    int X, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * third_lib/chamfer_dist/chamferdist  (pybind surface: chamferdist/ext.cpp:5-11)
 * K = 1, D = 3 specialisation of knn_points_idx / knn_points_backward
 * (knn.h:50-66, knn_cpu.cpp:7-58 & :64-106, knn.cu:297-435 & :482-544).
 *   p1 [N,P1,3] f32, p2 [N,P2,3] f32, lengths1/2 [N] i64 (device), idx [N,P1] i64,
 *   dist2 [N,P1] f32 = squared L2, evaluated as ((dx*dx+dy*dy)+dz*dz) without FMA; ties -> lowest
 *   index.  workspace: vidar_knn1_d3_workspace_bytes(N,P1) bytes of device scratch.
 * ------------------------------------------------------------------------- */
size_t vidar_knn1_d3_workspace_bytes(int N, int P1);
int vidar_knn1_d3_fwd(const float* p1, const float* p2, const int64_t* lengths1,
                      const int64_t* lengths2, int64_t* idx, float* dist2, void* workspace, int N,
                      int P1, int P2, void* stream);
/* grad_p1 [N,P1,3] written, grad_p2 [N,P2,3] zeroed then accumulated (fp32 atomics). */
int vidar_knn1_d3_bwd(const float* p1, const float* p2, const int64_t* lengths1,
                      const int64_t* lengths2, const int64_t* idx, const float* grad_dist2,
                      float* grad_p1, float* grad_p2, int N, int P1, int P2, void* stream);

/* ---------------------------------------------------------------------------
 * Multi-scale deformable attention.  Replaces mmcv._ext.ms_deform_attn_{forward,backward}
 * (mmcv-full 1.4.0, third party) as called from
 * projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:118-124, :150-160.
 *   value [B,Nv,H,C] f32 (C must be 32), spatial_shapes [L,2] i64 (h,w), level_start_index [L] i64,
 *   sampling_loc [B,Nq,H,L,P,2] f32 in [0,1] (x,y), attn_weight [B,Nq,H,L,P] f32, out [B,Nq,H*C].
 * im2col_step of the reference API has no meaning here and is dropped at the Python layer.
 * Limits (VIDAR_ERR_BAD_ARG otherwise): C == 32, L <= 16, L*P <= 64, and ONE batch element of `value` smaller than
 * 4 GiB (Nv*H*C*4 < 2^32: corner lines are addressed with 32-bit byte offsets; a larger `value` tensor is processed
 * in several launches over batch elements, each with its own base pointers -- same results).
 * A sample outside its level, or with a NaN location, contributes nothing (zero output / gradients).
 * bwd: grad_value is zeroed by the call then accumulated with fp32 atomics; grad_sampling_loc and
 * grad_attn_weight are fully written (the reference expects pre-zeroed buffers, function.py:146-148).
 * ------------------------------------------------------------------------- */
int vidar_msda_fwd_f32(const float* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, float* out, int B, int Nv, int H, int C, int Nq,
                       int L, int P, void* stream);
/* bwd has two scatter strategies for grad_value, chosen by the caller through `workspace`:
 *   workspace == NULL : one fp32 atomic per (sample, corner) 128-byte line (fine for small launches);
 *   workspace != NULL : samples are counting-sorted by destination tile (batch, level, 8x8 pixels, head)
 *                       into the workspace, accumulated per tile in LDS and flushed with one atomic per
 *                       non-zero window line (see csrc/msda.hip).  The workspace needs
 *                       vidar_msda_bwd_workspace_bytes(...) bytes (0 = shape not supported by this
 *                       strategy: >= 2^31 samples or grad_out of 2 GiB and more), is scratch (no state survives the
 *                       call) and must stay alive until the stream has run the call.
 * Both give the same result up to fp32 summation order. */
size_t vidar_msda_bwd_workspace_bytes(int B, int Nv, int H, int Nq, int L, int P);
/* tuning/A-B switch of the gather kernels (forward, grad_loc / grad_w): which 32 (batch, query, head) items a workgroup
 * owns.  1 (default) = head-major: 32 consecutive queries of ONE head, head = workgroup % H, so that with ViDAR's 8 heads
 * each of the 8 XCDs only reads its own head's plane of `value` (it fits the XCD's 4 MiB L2); 0 = 4 queries x 8 heads in
 * contiguous query bands per XCD (rounds 1-3).  Results do not depend on it.  Returns the previous value. */
int vidar_msda_set_item_order(int head_major);
int vidar_msda_bwd_f32(const float* value, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, const float* grad_out, float* grad_value,
                       float* grad_sampling_loc, float* grad_attn_weight, int B, int Nv, int H, int C,
                       int Nq, int L, int P, void* workspace, size_t workspace_bytes, void* stream);

/* Fused operand preparation (csrc/msda.hip): the op consumes the raw outputs of the sampling_offsets and
 * attention_weights Linear layers instead of prepared locations / weights, replacing the softmax, the
 * division by the level size, the reference-point add and the (TSA) permute copies of
 * temporal_self_attention.py:218-245, spatial_cross_attention.py:359-383, vidar_decoder.py:463-490.
 *   off_raw [bs,Nq,H,Qn,L,P,2] f32, logit_raw [bs,Nq,H,Qn,L*P] f32, ref [bs*Qn,Nq,R,2] f32,
 *   value [bs*Qn,Nv,H,C]; loc = ref[.., r, :] + off / (W_l, H_l) with r = level (mode 0, R == L) or
 *   point % R (mode 1, P % R == 0); w = softmax over L*P.
 * fwd writes out [bs*Qn,Nq,H*C] and the prepared operands loc_out [bs*Qn,Nq,H,L,P,2] / w_out [bs*Qn,Nq,H,L,P]
 * (what the backward needs); bwd takes those and writes grad_value (zeroed + accumulated),
 * grad_off_raw, grad_logit_raw (fully written, raw layouts).  workspace as for vidar_msda_bwd_f32 with
 * B = bs*Qn.
 * merge_queue = 1: the op also takes the MEAN over the Qn queue entries that TemporalSelfAttention applies to its output
 * (`output.view(bs, Qn, Nq, C).mean(1)`, temporal_self_attention.py:256-264): out / grad_out are [bs,Nq,H*C], the Qn
 * entries of a (batch, query, head) are gathered by one item and summed in registers (Qn*L*P <= 64); loc_out / w_out and
 * the gradients keep the layouts above.  merge_queue = 0: out / grad_out [bs*Qn,Nq,H*C]. */
int vidar_msda_fused_fwd_f32(const float* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index, const float* off_raw, const float* logit_raw,
                             const float* ref, float* loc_out, float* w_out, float* out, int bs, int Qn, int Nv,
                             int H, int C, int Nq, int L, int P, int R, int mode, int merge_queue, void* stream);
int vidar_msda_fused_bwd_f32(const float* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index, const float* sampling_loc,
                             const float* attn_weight, const float* grad_out, float* grad_value,
                             float* grad_off_raw, float* grad_logit_raw, int bs, int Qn, int Nv, int H, int C,
                             int Nq, int L, int P, int merge_queue, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * y = LayerNorm(dropout(x) + residual) and its backward in one pass each (csrc/norm_fuse.hip): the tail of every
 * attention / FFN block of the encoder and the future decoder (temporal_self_attention.py:270-271,
 * spatial_cross_attention.py:172-174, vidar_decoder.py:515-516, mmcv FFN) followed by the layer's norm.
 *   x, residual, y, sum_out [rows, C] f32 with C == 256; gamma, beta [C]; mean_out, rstd_out [rows].
 *   dropout keeps element i iff hash(seed, i) >= p (recomputed in the backward from the same seed; p = 0: exact).
 *   bwd: grad_x, grad_residual, grad_gamma, grad_beta fully written (the affine gradients from per-workgroup
 *   partial rows kept in `workspace` (vidar_drop_add_ln_bwd_workspace_bytes, scratch).
 * ------------------------------------------------------------------------- */
int vidar_drop_add_ln_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                              float* sum_out, float* mean_out, float* rstd_out, int64_t rows, int C, float p, float eps,
                              uint32_t seed, void* stream);
int vidar_drop_add_ln_bwd_f32(const float* grad_y, const float* sum_in, const float* gamma, const float* mean_in,
                              const float* rstd_in, float* grad_x, float* grad_residual, float* grad_gamma,
                              float* grad_beta, void* workspace, int64_t rows, int C, float p, uint32_t seed,
                              void* stream);
size_t vidar_drop_add_ln_bwd_workspace_bytes(int64_t rows); /* scratch for the per-workgroup affine-gradient partials */

/* y = dropout(relu(x)) and its backward, one pass each (csrc/norm_fuse.hip): the hidden activation of the FFN blocks
 * (mmcv FFN [3P]: Linear -> ReLU -> Dropout -> Linear; custom_base_transformer_layer.py builds them from the configs'
 * ffn_cfgs).  n elements (a multiple of 4), 0 <= p < 1; keeps element i iff hash(seed, i) >= p and scales it by
 * 1 / (1 - p) (same hash as the fused LayerNorm tail; not torch's Philox stream: dropout noise has no parity contract,
 * p = 0 is exact).  bwd: grad_x = y > 0 ? grad_y / (1 - p) : 0 -- y is the forward's OUTPUT, no mask is stored.
 * x / y and grad_y / grad_x may alias. */
int vidar_relu_drop_fwd_f32(const float* x, float* y, int64_t n, float p, uint32_t seed, void* stream);
int vidar_relu_drop_bwd_f32(const float* grad_y, const float* y, float* grad_x, int64_t n, float p, void* stream);

/* Column sums of a row-major [rows, cols] fp32 matrix: out[c] = sum_r x[r, c] -- the bias gradient of the Linear
 * layers on the path (what autograd computes with a generic `sum(0)` for every nn.Linear the reference's modules
 * own, e.g. temporal_self_attention.py:98-103, spatial_cross_attention.py:66, :244-248, vidar_decoder.py:358-363; mmcv FFN [3P]).  `out` [cols] is
 * zeroed and fully written by the call.  cols must be a multiple of 4 whose quarter is a power of two (<= 4096). */
int vidar_colsum_f32(const float* x, float* out, int64_t rows, int cols, void* stream);

/* ---------------------------------------------------------------------------
 * BEV-encoder bookkeeping for F frames at once.  Replaces BEVFormerEncoder.point_sampling
 * (projects/mmdet3d_plugin/bevformer/modules/encoder.py:96-156) and the visible-query rebatch index of
 * SpatialCrossAttention.forward (spatial_cross_attention.py:136-152, :164-171).
 *   ref_3d [D,Q,3] f32 pillar anchors in [0,1] (encoder.py:54-80), lidar2img [F,B,N,4,4] f32 row-major,
 *   pc_range: HOST pointer to 6 floats, img_h/img_w: padded image shape of camera 0 (encoder.py:137-138).
 * Outputs (caller-allocated, fully written): ref_cam [F,N,B,Q,D,2] f32, bev_mask [F,N,B,Q,D] u8 (0/1),
 *   count [F,B,Q] f32 = max(1, #cameras seeing the query), idx [F,N,Q] i64 = visible queries of batch item 0
 *   in ascending order, then distinct in-range filler (slot number), valid [F,N,Q] u8 = slot < lens, lens [F,N] i32.
 *   slot_of [F,N,Q] i32 = inverse of idx (slot of a query in its camera's list, -1 = not visible).
 * vidar_sca_rows_f32 / vidar_sca_combine_f32: the rebatch / scatter-back of SpatialCrossAttention
 * (spatial_cross_attention.py:141-152, :164-171) and their backwards, as row gathers (csrc/bev_prep.hip):
 *   rows   : dst [B,N,S,C] = valid[n,s] ? src[b, idx[n,s], :] (/ count[b,idx] if count) : 0;  idx/valid rows have
 *            `stride` elements (S <= stride: a [N,S] slice of the [N,Q] tables);
 *   combine: dst [B,Q,C]   = sum over cameras n with slot_of[n,q] in [0,S) of src[b,n,slot_of[n,q],:] (/ count[b,q]).
 * C must be a multiple of 4. */
int vidar_sca_plan_f32(const float* ref_3d, const float* lidar2img, float* ref_cam, uint8_t* bev_mask,
                       float* count, int64_t* idx, uint8_t* valid, int32_t* lens, int32_t* slot_of,
                       const float* pc_range, float img_h, float img_w, int F, int B, int N, int Q, int D,
                       void* stream);
int vidar_sca_rows_f32(const float* src, const int64_t* idx, const uint8_t* valid, const float* count, float* dst,
                       int B, int N, int S, int stride, int Q, int C, void* stream);
int vidar_sca_combine_f32(const float* src, const int32_t* slot_of, const float* count, float* dst, int B, int N,
                          int S, int Q, int C, void* stream);

/* ---------------------------------------------------------------------------
 * LatentRendering ray-march (fused).  Replaces the torch op chain of
 * projects/mmdet3d_plugin/bevformer/modules/ray_operations/latent_rendering.py:96-150.
 * All maps are channel-last [bs, H*W, Z] f32 with Z == 16 (pred_height == embed_dims/reduction).
 * step = grid_step / (min(H,W)//2) rounded to f32 (:102-104); act: 0 = 'sigmoid', 1 = 'exp' (:117-123).
 *   prob   : occ logits -> path_prob = prod_{k<G,valid}(1 - act(occ(n_k))) * act(occ(n_cell))   (:96-129)
 *   gather : feat = sum_k a(n_k) m_k / (sum_k m_k + eps), m_k = path_prob(n_k)*valid2_k          (:131-150)
 * Backward entry points zero their gradient outputs and accumulate with fp32 atomics; `workspace`
 * (vidar_latent_render_bwd_workspace_bytes, caller-owned scratch, 16-byte aligned) holds 8 private copies of the
 * gradient maps as for the ray ops below, NULL = add straight into the outputs.
 * ------------------------------------------------------------------------- */
size_t vidar_latent_render_bwd_workspace_bytes(int bs, int H, int W, int Z, int maps /* 1: prob_bwd, 2: gather_bwd */);
int vidar_latent_render_prob_fwd_f32(const float* occ, float* path_prob, int bs, int H, int W, int Z,
                                     int grid_num, float step, int act, void* stream);
int vidar_latent_render_prob_bwd_f32(const float* occ, const float* grad_path_prob, float* grad_occ,
                                     int bs, int H, int W, int Z, int grid_num, float step, int act,
                                     void* workspace, size_t workspace_bytes, void* stream);
int vidar_latent_render_gather_fwd_f32(const float* path_prob, const float* lora_a, float* feat,
                                       float* msum, int bs, int H, int W, int Z, int grid_num,
                                       float step, float eps, void* stream);
int vidar_latent_render_gather_bwd_f32(const float* path_prob, const float* lora_a, const float* feat,
                                       const float* msum, const float* grad_feat,
                                       float* grad_path_prob, float* grad_lora_a, int bs, int H,
                                       int W, int Z, int grid_num, float step, float eps,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * ViDAR head ray-march over the predicted occupancy volume (fused).  Replaces the torch op chains
 * of projects/mmdet3d_plugin/bevformer/dense_heads/vidar_head_base.py:
 *   ray_ce     _get_grid_features :420-509 + cross_entropy(label 0) :586-592
 *   ray_gumbel dense rays :594-630 + _custom_gumbel_softmax_distance :754-773
 *   ray_argmax get_point_cloud_prediction :697-731 (test-time decode)
 * sigma [F,Z,Y,X] f32 logits; origin [F,3]; pts [R,3] ray end points; tindex [R] f32 frame slot
 * (<0 / NaN / >=F: ray skipped); all in voxel units.  K must be 512 (ray_grid_num of the released
 * configs); step = ray_grid_step.  Outputs are per ray; reductions stay in the host framework.
 *   ce[r]   = logsumexp_k(logit_k) - logit_0 over {end point, K waypoints}; valid[r] = 1 iff the
 *             end point lies strictly inside the volume (others are dropped, :464-467); lse saved.
 *   dist[r] = ((1-pn)+pn) * pd, pd = length of argmax_k(logit_k + noise[r,k]),
 *             pn = softmax mass of waypoints farther than pd;  aux[r] = {pd, pn, lse}.
 * Backward entry points zero grad_sigma and accumulate with fp32 atomics.  All rays of a frame start at the sensor
 * origin, so the first waypoints' atomics serialise on a few hundred addresses: given a `workspace` of
 * vidar_ray_bwd_workspace_bytes(F,Z,Y,X) bytes (caller-owned device scratch on the call's device / stream, nothing
 * survives the call) the kernels add into 8 private copies of the volume that a second kernel sums; with
 * workspace == NULL they add straight into grad_sigma.  Results agree to fp32 summation order.
 * ------------------------------------------------------------------------- */
size_t vidar_ray_bwd_workspace_bytes(int F, int Z, int Y, int X);
int vidar_ray_ce_fwd_f32(const float* sigma, const float* origin, const float* gt_pts,
                         const float* tindex, float* ce, float* lse, float* valid, int F, int R,
                         int Z, int Y, int X, int K, float step, void* stream);
int vidar_ray_ce_bwd_f32(const float* sigma, const float* origin, const float* gt_pts,
                         const float* tindex, const float* lse, const float* grad_ce,
                         float* grad_sigma, int F, int R, int Z, int Y, int X, int K, float step,
                         void* workspace, size_t workspace_bytes, void* stream);
int vidar_ray_gumbel_fwd_f32(const float* sigma, const float* origin, const float* pts,
                             const float* tindex, const float* noise /*[R,K]*/, float* dist,
                             float* aux /*[R,3]*/, int F, int R, int Z, int Y, int X, int K,
                             float step, void* stream);
int vidar_ray_gumbel_bwd_f32(const float* sigma, const float* origin, const float* pts,
                             const float* tindex, const float* aux, const float* grad_dist,
                             float* grad_sigma, int F, int R, int Z, int Y, int X, int K, float step,
                             void* workspace, size_t workspace_bytes, void* stream);
int vidar_ray_argmax_f32(const float* sigma, const float* origin, const float* pts,
                         const float* tindex, float* pred_dist, float* gt_dist, int F, int R, int Z,
                         int Y, int X, int K, float step, void* stream);

/* ---------------------------------------------------------------------------
 * Modulated deformable convolution v2 (backbone, "next" row SURVEY 8f-1).  Replaces the sampling
 * kernels of mmcv.ops.ModulatedDeformConv2dPack (mmcv-full 1.4.0, third party); the convolution
 * itself is a GEMM of weight [Cout, C*kh*kw] with the column matrix built here.
 *   x [N,C,H,W], offset [N,2*kh*kw,Ho,Wo] ((dy,dx) per tap), mask [N,kh*kw,Ho,Wo],
 *   cols / grad_cols [N, C*kh*kw, Ho*Wo].  deform_groups == 1.
 * col2im: grad_offset / grad_mask written; grad_x by one of two strategies chosen through `workspace`:
 *   NULL  : zeroed by the call, then one fp32 atomic per (channel, tap, pixel, corner);
 *   else  : the sampling positions are shared by all channels (deform_groups == 1), so the call sorts the
 *           4*K*P corner entries of every image by destination pixel into the workspace
 *           (vidar_dcn_col2im_workspace_bytes, scratch) and every destination pixel gathers its
 *           contributions -- no atomics on grad_x, fully written.  Same result up to fp32 summation order.
 * ------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------
 * Matrix products of the hot path on the gfx950 matrix cores (csrc/gemm_mfma.hip), fp32 in HBM on both sides:
 *     C[z] = act( (A[z] (M x K) * B[z] (K x N)) * scale + shift + residual[z] )      z = 0 .. batch-1
 * Replaces the library GEMMs behind the reference's nn.Linear layers -- first of all the attention value projection
 * (spatial_cross_attention.py:333-340 `value = self.value_proj(value)`, temporal_self_attention.py:197-208,
 * vidar_decoder.py:452-460), then sampling_offsets / attention_weights / output_proj of the same modules, the mmcv
 * FFN and the head MLPs -- and behind the 1x1 / deformable convolutions of the image backbone with the frozen
 * BatchNorm (+ residual + ReLU) that follows them folded into the epilogue (config vidar_1_8_nusc_1future.py:88-106).
 *   a_layout / b_layout: 0 = K-major (the contraction index is contiguous: element (m,k) at A[m*lda + k], element
 *     (k,n) at B[n*ldb + k], i.e. an nn.Linear weight [N,K] as B), 1 = MN-major (element (m,k) at A[k*lda + m],
 *     element (k,n) at B[k*ldb + n], i.e. an NCHW activation [C, H*W] as B).  C is row-major, C[m*ldc + n].
 *     Any alignment of the row starts is accepted (4-byte), any M, N, K >= 1.
 *   strideA/B/C/R: element offsets between batch items (0 = shared operand).
 *   scale, shift: optional vectors indexed by n (vec_axis 0: the bias of F.linear) or by m (vec_axis 1: per output
 *     channel of a convolution); residual: optional [M, N] tensor with leading dimension ldr; relu: 0/1.
 *   precision: 0 = VIDAR_GEMM_F32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -- the arithmetic
 *     of the library kernels it replaces); 1 = VIDAR_GEMM_BF16X3: every operand element is split x = hi + lo into
 *     two bf16 while it is staged (hi = rne(x), lo = rne(x - hi): 16 significand bits, relative error <= 2^-16) and
 *     a product is lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The reference runs
 *     these products in TF32 (11 significand bits; torch 1.10.1+cu111 defaults, README.md:96; tools/train.py:141-144
 *     only disables it under `close_tf32`, which no ViDAR config sets; encoder.py:98-100 is the one opt-out and
 *     stays fp32 here too).
 *   reduce = 1: ONE output C = act(sum_z A[z]*B[z] ...) -- the batch items are summed and K is additionally split
 *     (vidar_gemm_splits) so that a weight gradient fills the chip; partial products go to `workspace`
 *     (vidar_gemm_workspace_bytes) as fp32 slabs and are summed in a fixed order: deterministic, no atomics.
 *   a_rowsum (reduce = 1 and an MN-major A only; else NULL): a_rowsum[m] = sum_z sum_k A[z][m, k] written next to C --
 *     with A = grad_out^T this is the bias gradient of the nn.Linear whose weight gradient the call computes
 *     (autograd's grad_out.sum(0)): the tile-column-0 workgroups add up the A values they stage anyway, partial rows
 *     go behind the slabs and are reduced in the same fixed order.  Deterministic, and no second pass over grad_out.
 * ------------------------------------------------------------------------- */
#define VIDAR_GEMM_F32 0
#define VIDAR_GEMM_BF16X3 1
#define VIDAR_GEMM_K_MAJOR 0
#define VIDAR_GEMM_MN_MAJOR 1
int vidar_gemm_f32(const float* A, int64_t lda, int a_layout, const float* B, int64_t ldb, int b_layout, float* C,
                   int64_t ldc, int M, int N, int K, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                   const float* scale, const float* shift, int vec_axis, const float* residual, int64_t ldr,
                   int64_t strideR, int relu, int precision, int reduce, float* a_rowsum, void* workspace,
                   size_t workspace_bytes, void* stream);
int vidar_gemm_splits(int M, int N, int K, int batch, int precision, int reduce);
/* scratch of a reduced product: reduce = 1 the product alone (0 bytes when it is a single slab), reduce = 2 the product
 * with the row sums of A (`a_rowsum`): slabs + one partial row per slab */
size_t vidar_gemm_workspace_bytes(int M, int N, int K, int batch, int precision, int reduce);
/* A/B switch of the kernel's structure: 0 (default) = every wave does every job; 1 / 2 = wave-specialised workgroups
 * (4 staging waves + 4 matrix / epilogue waves, LDS double buffer, one LDS-only barrier per k-step), one / two per CU
 * (csrc/gemm_mfma.hip; measured equal or slower: profiles/r05_kbench_gemm_variants.log).  Same arithmetic, same tile
 * order: results are bit-identical.  Returns the previous value. */
int vidar_gemm_set_variant(int variant);

/* A/B switch of the DCNv2 sampling kernels.  bit 0 (default on): the grad_x gather of vidar_dcn_col2im_f32 (workspace form)
 * on 3x3 / stride 1 / dilation 1 layers copies, per tap, the window of grad_cols that can reach an 8 x 32 tile of
 * destination pixels (+ 4 pixels of learned offset) into LDS and gathers from there (sources beyond the window: global
 * loads); 0 = 4-byte global gathers everywhere (rounds 2-5).  Same sums up to fp32 order.  Values outside 0..1 are ignored.
 * Returns the previous value. */
int vidar_dcn_set_variant(int variant);
int vidar_dcn_im2col_f32(const float* x, const float* offset, const float* mask, float* cols, int N,
                         int C, int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad,
                         int dil, void* stream);
int vidar_dcn_col2im_f32(const float* grad_cols, const float* x, const float* offset,
                         const float* mask, float* grad_x, float* grad_offset, float* grad_mask,
                         int N, int C, int H, int W, int Ho, int Wo, int kh, int kw, int stride,
                         int pad, int dil, void* workspace, size_t workspace_bytes, void* stream);
size_t vidar_dcn_col2im_workspace_bytes(int N, int H, int W, int Ho, int Wo, int kh, int kw);

/* 3x3 convolution (stride 1, pad 1, dilation 1, groups 1) with FEW output channels as an implicit GEMM on the fp32
 * matrix cores:  out[N,Cout,H,W] = conv2d(x[N,C,H,W], weight[Cout,C,3,3]) + bias  (bias may be NULL).
 * Replaces the library convolution behind `conv_offset` of mmcv's ModulatedDeformConv2dPack -- the 3x3 that predicts the
 * 18 offsets + 9 masks of every DCNv2 bottleneck (vidar_1_8_nusc_1future.py:93-95: DCNv2 on ResNet101 stages 3 and 4 =
 * 26 of the backbone's 37 3x3 convolutions); its backward stays the library's.  fp32 products are exact, the sum runs
 * channel-major with the 9 taps inside (within 1e-5 of an fp64 convolution, tests/test_conv3x3_gpu.py).
 * workspace: vidar_conv3x3_few_workspace_bytes(C) bytes (the packed weights; written by the call).
 * Limits (VIDAR_ERR_BAD_ARG otherwise): Cout <= 32, C % 8 == 0, W <= 191, H*W < 2^30. */
size_t vidar_conv3x3_few_workspace_bytes(int C);
int vidar_conv3x3_few_f32(const float* x, const float* weight, const float* bias, float* out, int N, int C, int H, int W,
                          int Cout, void* workspace, size_t workspace_bytes, void* stream);

/* Fused frozen-BatchNorm epilogue of the backbone: y = act(x*scale[c] + shift[c] (+ residual)),
 * NCHW, scale = gamma/sqrt(var+eps), shift = beta - mean*scale (BN is frozen / eval in every ViDAR
 * config: vidar_1_8_nusc_1future.py:93-95).  relu: 0/1.  residual / grad_residual may be NULL.
 * bwd needs y (the forward output) for the ReLU mask; scale/shift receive no gradient (frozen). */
int vidar_affine_act_fwd_f32(const float* x, const float* scale, const float* shift,
                             const float* residual, float* y, int N, int C, int HW, int relu,
                             void* stream);
/* Stem of the backbone in one pass: y[N,C,Ho,Wo] = max_pool2d(relu(x*scale[c] + shift[c]), 3, stride 2, pad 1) with
 * Ho = (H-1)/2 + 1, Wo = W/2 -- mmdet ResNet.forward's conv1 -> norm1 -> relu -> maxpool with the frozen BN of the
 * ViDAR configs (vidar_1_8_nusc_1future.py:86-95, frozen_stages=1: forward only).  Bit-identical to
 * vidar_affine_act_fwd_f32(relu=1) followed by the pooling.  VIDAR_ERR_BAD_ARG unless W % 4 == 0, x 16-byte and
 * y 8-byte aligned, N*C <= 65535 (the caller then takes the two-kernel path). */
int vidar_stem_bn_relu_pool_f32(const float* x, const float* scale, const float* shift, float* y, int N, int C,
                                int H, int W, void* stream);
int vidar_affine_act_bwd_f32(const float* grad_y, const float* y, const float* scale, float* grad_x,
                             float* grad_residual, int N, int C, int HW, int relu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDAR_HIP_H_ */
