/* dmnerf_hip.h -- C ABI of the MI355X-native (gfx950) DM-NeRF volumetric-render hot path.
 *
 * The reference (ChenLu-china/DM-NeRF) is pure Python; it has no FFI layer.  The drop-in
 * boundary is therefore its Python function API.  This header is the
 * C ABI beneath the Python shim `dm_nerf_amd`: plain pointers and sizes, no torch types.
 * Each entry point names the reference interface (file:line under the reference root) whose
 * computation it replaces.
 *
 * Conventions
 *   - all device pointers are fp32 (or int32 where stated), dense, row-major, 16-byte aligned
 *     allocations; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - functions enqueue work on `stream` and return immediately; they never allocate device
 *     memory and never synchronise (safe to capture in a hipGraph);
 *   - return value: DMN_OK (0) or a negative DMN_ERR_* code; `dmn_error_string` explains it;
 *   - `z` depth arrays take a ray stride: element (n, s) is z[n * z_ray_stride + s];
 *     z_ray_stride = 0 means one depth lattice shared by all rays (the expanded view returned
 *     by z_val_sample, networks/helpers.py:119).
 */
#ifndef DMNERF_HIP_H_
#define DMNERF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMN_OK 0
#define DMN_ERR_BAD_ARG (-1)      /* null pointer, non-positive size, unsupported shape */
#define DMN_ERR_UNSUPPORTED (-2)  /* e.g. ins_num + 1 > 128, S too large for the LDS path */
#define DMN_ERR_LAUNCH (-3)       /* hipGetLastError() after a launch was not hipSuccess */
#define DMN_ERR_NO_DEVICE (-4)

/* arithmetic of the field kernel */
#define DMN_PREC_FP32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 products (parity mode) */
#define DMN_PREC_FP16X3 1  /* 3 x v_mfma_f32_32x32x16_f16 per product on hi/lo fp16 splits (~1e-6 rel.) */
#define DMN_PREC_BF16X3 2  /* 3 x v_mfma_f32_32x32x16_bf16 per product on hi/lo bf16 splits (~1e-5 rel.) */
#define DMN_PREC_BF16X6 3  /* 6 x v_mfma_f32_32x32x16_bf16 per product on EXACT hi/mid/lo bf16 splits of both operands:
                              all 24 significand bits of the fp32 operands enter the products (fp32-grade arithmetic) */
#define DMN_PREC_BF16X6W 4 /* the DMN_PREC_BF16X6 arithmetic on v_mfma_f32_16x16x32_bf16 (csrc/field_fwd_x6w.hip): inference
                            * entry points only (dmn_render_rays), "x6w" blob; another summation order,
                            * so results differ from DMN_PREC_BF16X6 by fp32 rounding */
/* bits of the sticky status word the split-precision kernels OR into */
#define DMN_STATUS_FP16_RANGE 1  /* an activation left the fp16 range: re-run with another precision */

int dmn_abi_version(void);
const char* dmn_error_string(int code);
/* Writes the name of device 0 (e.g. "gfx950:...") into buf; DMN_ERR_NO_DEVICE without a GPU. */
int dmn_device_name(char* buf, int buf_len);

/* ---- packed weights ---------------------------------------------------------------------
 * One field network (DM_NeRF, networks/dm_nerf.py:58-106) = `dmn_field_param_floats(ins_num)`
 * fp32 values in the reference's state_dict order (mlps.0.weight, mlps.0.bias, ...,
 * rgb_linear.bias; config.py:146-159 builds two such networks).  `dmn_field_pack` gathers
 * them into the kernel's blob (biases + MFMA-ordered weight stream, csrc/field_layout.h).    */
int dmn_field_param_floats(int ins_num);
long dmn_field_blob_floats(int ins_num);
/* host: index_out[blob_floats] <- index into the flat parameter vector, or -1 (zero pad) */
int dmn_field_pack_index(int ins_num, int32_t* index_out);
/* device: blob[i] = index[i] < 0 ? 0 : params[index[i]] */
int dmn_field_pack(const float* params, const int32_t* index, long blob_floats, float* blob, void* stream);

/* ---- get_rays_k (networks/helpers.py:50-61) --------------------------------------------
 * rays_o, rays_d: [H*W, 3].  K is passed as the five entries the reference reads
 * (K[0][0], K[0][2], K[1][1], K[1][2], K[2][2], helpers.py:55) rounded to fp32, c2w as the
 * 12 host floats of its top 3x4 block (helpers.py:58,60).                                   */
int dmn_raygen(int H, int W, float k00, float k02, float k11, float k12, float k22, const float* c2w_3x4_host,
               float* rays_o, float* rays_d, void* stream);

/* Rays of n selected pixels (flat index row * W + col, int64, device) -- the ray part of get_select_full /
 * get_select_crop (networks/helpers.py:64-111), without generating the other H*W - n rays.            */
int dmn_raygen_select(int H, int W, float k00, float k02, float k11, float k12, float k22, const float* c2w_3x4_host,
                      const long* pixel_index, long n, float* rays_o, float* rays_d, void* stream);

/* ---- stratified jitter (networks/render.py:40-47) --------------------------------------- */
int dmn_jitter_depths(const float* z, long z_ray_stride, const float* t_rand, long n_rays, int n_samples,
                      float* z_out, void* stream);

/* ---- Embedder.embed (networks/dm_nerf.py:37-38; bands :25) ------------------------------
 * x: [M,3] -> out: [M, 3 + 6*n_freqs], ordering of dm_nerf.py:17-32.                         */
int dmn_encode(const float* x, long n_points, int n_freqs, float* out, void* stream);

/* ---- field evaluation: points -> encodings -> DM_NeRF.forward ----------------------------
 * Replaces networks/render.py:37-38,49-61 (and :71-83 for the fine pass): viewdirs =
 * d/|d|, pts = o + d*z, both encodings, and the whole network (networks/dm_nerf.py:80-106),
 * fused in one kernel (fp32 MFMA).  raw: [n_rays * n_samples, 4 + ins_num + 1] laid out
 * [rgb(3), sigma(1), ins(ins_num+1)] (dm_nerf.py:105).                                       */
int dmn_field_fwd_rays(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                       int n_samples, const float* blob, int ins_num, float* raw, void* stream);
/* DM_NeRF.forward on caller-made encodings (networks/manipulator.py:125-131,
 * tools/mesh_generator.py:41-45): embedded [M, 90] -> raw [M, 4 + ins_num + 1].              */
int dmn_field_fwd_embedded(const float* embedded, long n_points, const float* blob, int ins_num, float* raw,
                           void* stream);

/* Split-precision flavours of the two calls above (same reference lines, same outputs).  `blob` is
 * the x3 blob (dmn_field_blob_x3_bytes / dmn_field_pack_index_x3 / dmn_field_pack_x3, made for the same
 * `precision`), `status` an optional device int the kernel ORs DMN_STATUS_* bits into.               */
long dmn_field_blob_x3_bytes(int ins_num);
long dmn_field_pack_index_x3_len(int ins_num);
int dmn_field_pack_index_x3(int ins_num, int32_t* index_out);
int dmn_field_pack_x3(const float* params, const int32_t* index, int precision, void* blob, void* stream);
/* the same gather for any x3-format table of `index_len` entries (forward or backward-data blob) */
int dmn_field_pack_x3_n(const float* params, const int32_t* index, long index_len, int precision, void* blob,
                        void* stream);
int dmn_field_fwd_rays_x3(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                          int n_samples, const void* blob, int ins_num, int precision, float* raw, int* status,
                          void* stream);
int dmn_field_fwd_embedded_x3(const float* embedded, long n_points, const void* blob, int ins_num, int precision,
                              float* raw, int* status, void* stream);
/* Density only: the trunk and density_linear (dm_nerf.py:83-87,101), raw[..., 3] = sigma and every other channel 0.
 * For the coarse pass of an image loop, whose colour / instance outputs nobody reads (networks/tester.py:70-76 keeps
 * rgb_fine and ins_fine): the compositing weights that drive the resampler depend on sigma alone (render.py:16-18). */
int dmn_field_fwd_rays_x3_sigma(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                long n_rays, int n_samples, const void* blob, int ins_num, int precision, float* raw,
                                int* status, void* stream);

/* Full-precision flavour on the bf16 matrix pipe (DMN_PREC_BF16X6; csrc/field_fwd_x6.hip): same reference lines and
 * outputs as dmn_field_fwd_rays / dmn_field_fwd_embedded.  Every fp32 operand is split exactly into three bf16 values
 * (8 + 8 + 8 significand bits) and the product is the sum of the six partial products above 2^-24 of it, accumulated in
 * fp32 -- no operand bit is dropped, so the deviation from the reference is the rounding noise of fp32 accumulation, as
 * for the exact-fp32 MFMA kernel (tests/test_gpu_parity.py measures both against an fp64 evaluation).  `blob` is the x6
 * blob (three bf16 planes per weight: dmn_field_blob_x6_bytes / dmn_field_pack_index_x6 / dmn_field_pack_x6).
 * sigma_only != 0: the trunk and density_linear only, as dmn_field_fwd_rays_x3_sigma.                              */
long dmn_field_blob_x6_bytes(int ins_num);
long dmn_field_pack_index_x6_len(int ins_num);
int dmn_field_pack_index_x6(int ins_num, int32_t* index_out);
int dmn_field_pack_x6(const float* params, const int32_t* index, int ins_num, void* blob, void* stream);
/* The same arithmetic on v_mfma_f32_16x16x32_bf16 (csrc/field_fwd_x6w.hip; inference): same blob size and gather
 * kernel (dmn_field_blob_x6_bytes, dmn_field_pack_index_x6_len, dmn_field_pack_x6), its own element order
 * (dmn_field_pack_index_x6w); replaces the same reference lines as dmn_field_fwd_rays_x6 / _embedded_x6.  Every head
 * width the x6 entry points serve (ins_num + 1 <= 128). */
int dmn_field_pack_index_x6w(int ins_num, int32_t* index_out);
int dmn_field_fwd_rays_x6w(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                           int n_samples, const void* blob, int ins_num, int sigma_only, float* raw, void* stream);
int dmn_field_fwd_embedded_x6w(const float* embedded, long n_points, const void* blob, int ins_num, float* raw,
                               void* stream);
int dmn_field_fwd_rays_x6(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                          int n_samples, const void* blob, int ins_num, int sigma_only, float* raw, void* stream);
int dmn_field_fwd_embedded_x6(const float* embedded, long n_points, const void* blob, int ins_num, float* raw,
                              void* stream);
/* The same with the save record of dmn_field_fwd_rays_save (training: the backward kernels read it; the record does not
 * depend on which forward wrote it).                                                                                */
int dmn_field_fwd_rays_x6_save(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                               int n_samples, const void* blob, int ins_num, float* raw, float* save, void* stream);

/* ---- field evaluation + compositing (+ resampling) fused: `raw` never reaches memory ----------------------
 * For the image loop (networks/tester.py:62-76 keeps rgb_fine / ins_fine only).  One launch evaluates the field of
 * `precision` (DMN_PREC_FP16X3 / DMN_PREC_BF16X3 with the x3 blob, DMN_PREC_BF16X6 with the x6 blob) on o + d*z, composites every ray as
 * dmn_composite_fwd does (render_train, networks/render.py:6-28; keep_last = 0) and, when z_fine != NULL, runs the
 * resampling of dmn_resample_merge on the ray's weights (networks/render.py:66-70: draws u as there, NULL = det).
 * rgb [N,3], depth [N], ins [N,ins_num], weights [N,S] may each be NULL.  sigma_only: the network stops behind its
 * density head (colour / instance channels composite as zeros).  n_samples must be a multiple of 64 (a workgroup walks
 * two rays in 64-sample passes), else DMN_ERR_UNSUPPORTED.  Bit-identical to the unfused entry points.             */
int dmn_field_composite_x3(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                           int n_samples, const void* blob, int ins_num, int precision, int sigma_only, float* rgb,
                           float* depth, float* ins, float* weights, const float* u, long u_ray_stride, int n_new,
                           float* z_fine, int* status, void* stream);

/* The inference chunk of dm_nerf in two such launches: coarse field + compositing + resampling -> z_fine
 * [N, n_samples + n_importance] (caller-owned; z_vals_fine of networks/render.py:70), fine field + compositing ->
 * rgb_fine / depth_fine / ins_fine.  Coarse maps may be NULL; coarse_sigma_only as above.                          */
int dmn_render_fused(const float* rays_o, const float* rays_d, const float* z_coarse, long z_ray_stride,
                     const float* u, long u_ray_stride, long n_rays, int n_samples, int n_importance,
                     const void* blob_coarse, const void* blob_fine, int ins_num, int precision,
                     int coarse_sigma_only, float* z_fine, float* rgb_fine, float* depth_fine, float* ins_fine,
                     float* rgb_coarse, float* depth_coarse, float* ins_coarse, int* status, void* stream);

/* ---- the whole chunk in one call: dm_nerf (networks/render.py:31-96) ------------------------------------
 * (stratified jitter with t_rand, or none when NULL ->) coarse field -> compositing -> inverse-CDF resampling
 * (draws u, or the det = True lattice when NULL) + sorted merge -> fine field -> compositing, all enqueued on
 * `stream`.  z_coarse is the caller's depth lattice as in the reference's signature (z_vals_coarse).  blobs are
 * the packed weights of `precision` (dmn_field_pack / _x3 / _x6).  workspace: dmn_render_workspace_floats()
 * floats laid out [z_coarse n*S | raw_coarse n*S*C | weights_coarse n*S | z_vals_fine n*(S+Ni) | raw_fine n*(S+Ni)*C],
 * C = 5 + ins_num -- the raw_* / z_vals_* entries of the reference's result dict.  Coarse outputs may be NULL.   */
long dmn_render_workspace_floats(long n_rays, int n_samples, int n_importance, int ins_num);
int dmn_render_rays(const float* rays_o, const float* rays_d, const float* z_coarse, long z_ray_stride,
                    const float* t_rand, const float* u, long u_ray_stride, long n_rays, int n_samples,
                    int n_importance, const void* blob_coarse, const void* blob_fine, int ins_num, int precision,
                    float* workspace, float* rgb_fine, float* depth_fine, float* ins_fine, float* rgb_coarse,
                    float* depth_coarse, float* ins_coarse, int* status, void* stream);

/* ---- object manipulation: label surgery between an original and a transformed scene (networks/manipulator.py:18-83)
 * dmn_sigmoid_argmax: labels[i] = argmax_c sigmoid(x[i*row_stride + c]), c < n_channels, first index among equals
 *   (torch.argmax(torch.sigmoid(.), -1); per sample point on raw + 4 with n_channels = C - 4, per ray on the
 *   composited instance map without its last channel).
 * dmn_exchange: one moved object.  ori_raw [N,S,C] is edited in place (rows taken from tar_raw, or emptied),
 *   ori_labels [N,S] is updated in place (occluders inherit the ray's label), tar_labels [N,S] is written.       */
int dmn_sigmoid_argmax(const float* x, long rows, long row_stride, int n_channels, int* labels, void* stream);
int dmn_exchange(float* ori_raw, const float* tar_raw, int* ori_labels, const int* ori_ray_labels,
                 const int* tar_ray_labels, int* tar_labels, long n_rays, int n_samples, int channels, int move_label,
                 void* stream);

/* ---- training: backward of the field (autograd of networks/dm_nerf.py:80-106 inside train_dmsr.py:32-64) ----
 * The forward writes a save record (register images per 32-point tile: encodings, every layer's output, ReLU
 * sign bits), the backward-data call turns dL/d raw [M,C] into dL/d pre-activation of every layer (gradient
 * record, same image layout; the instance branch stops at the trunk output, dm_nerf.py:95), and the
 * weight-gradient call reduces gradient x input over the points into the flat parameter-gradient vector
 * (state_dict order, dmn_field_param_floats entries).  Forward and backward-data are exact fp32 MFMA; the
 * weight-gradient GEMMs run in DMN_PREC_FP32, DMN_PREC_BF16X6 (six bf16 MFMAs per product on exact three-way splits:
 * fp32-grade, the default of the fp32-grade forwards) or DMN_PREC_BF16X3 (three bf16 MFMAs per product, 16 operand
 * bits; opt-in), fp32 accumulate; DMN_PREC_FP16X3 is refused: gradients underflow fp16).
 *   dmn_field_save_floats / dmn_field_grad_floats: buffer sizes for n_points
 *   dmn_field_bwd_blob_floats / dmn_field_bwd_pack_index: transposed-weight blob, packed with dmn_field_pack
 *   dmn_field_wgrad_index_len / dmn_field_wgrad_index: scatter table (upload once per ins_num)
 *   dmn_field_wgrad_workspace_floats: partial sums, `splits` slices of the points (a few per CU)        */
long dmn_field_save_floats(long n_points);
long dmn_field_grad_floats(long n_points, int ins_num);
int dmn_field_fwd_rays_save(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                            int n_samples, const float* blob, int ins_num, float* raw, float* save, void* stream);
/* the same with the split-precision forward (x3 blob); DMN_PREC_BF16X3 only: there is no place to consult the fp16
 * range flag between forward and backward */
int dmn_field_fwd_rays_x3_save(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride, long n_rays,
                               int n_samples, const void* blob, int ins_num, int precision, float* raw, float* save,
                               void* stream);
long dmn_field_bwd_blob_floats(int ins_num);
int dmn_field_bwd_pack_index(int ins_num, int32_t* index_out);
/* split-precision flavour of the backward-data step (bf16, three MFMA products per fp32 product): x3-format
 * transposed-weight blob (pack with dmn_field_pack_x3_n and DMN_PREC_BF16X3) and the kernel */
long dmn_field_bwd_blob_x3_bytes(int ins_num);
long dmn_field_bwd_pack_index_x3_len(int ins_num);
int dmn_field_bwd_pack_index_x3(int ins_num, int32_t* index_out);
int dmn_field_bwd_data_x3(const float* g_raw, const float* save, const void* bwd_blob_x3, long n_points, int ins_num,
                          float* grad_record, void* stream);
/* full-precision bf16 flavour of the backward-data step (DMN_PREC_BF16X6: six bf16 MFMA products per fp32 product on
 * exact three-way splits of both operands, csrc/field_bwd_x6.hip) -- the reference back-propagates in fp32
 * (train_dmsr.py:62-64 through networks/dm_nerf.py:80-106, h.detach() at :95): x6-format transposed-weight blob
 * (dmn_field_bwd_pack_index_x6 + dmn_field_pack_x6_n) and the kernel; same save / gradient records as the others */
long dmn_field_bwd_blob_x6_bytes(int ins_num);
long dmn_field_bwd_pack_index_x6_len(int ins_num);
int dmn_field_bwd_pack_index_x6(int ins_num, int32_t* index_out);
int dmn_field_pack_x6_n(const float* params, const int32_t* index, long index_len, void* blob, void* stream);
int dmn_field_bwd_data_x6(const float* g_raw, const float* save, const void* bwd_blob_x6, long n_points, int ins_num,
                          float* grad_record, void* stream);
long dmn_field_wgrad_index_len(int ins_num);
int dmn_field_wgrad_index(int ins_num, int32_t* index_out);
int dmn_field_bwd_data(const float* g_raw, const float* save, const float* bwd_blob, long n_points, int ins_num,
                       float* grad_record, void* stream);
long dmn_field_wgrad_workspace_floats(int ins_num, int splits);
/* With DMN_PREC_BF16X3 the smallest GEMMs run on a second, library-owned stream beside the kernel of the nine 256x256
 * GEMMs (forked from and joined to `stream` by events: everything the call launches is ordered after the work already
 * on `stream` and before the work submitted to it afterwards); that stream and its two events are created once per
 * device on first use -- the only objects this library ever creates. */
int dmn_field_bwd_weights(const float* save, const float* grad_record, long n_points, int ins_num,
                          const int32_t* wgrad_index, int splits, int precision, float* workspace,
                          float* grads_flat, int accumulate, void* stream);

/* ---- emptiness penalizer (networks/penalizer.py:5-62: ins_penalizer -> emptiness_penalizer) ------------------
 * raw [N,S,C] (logits = channels 4..C-1), z (stride), depth [N] (the rendered depth, detached by the caller as
 * penalizer.py:59 does), rays_d [N,3].  Forward fills sums4 (device doubles): sum of the "before" term, number of
 * samples before the tolerance band, sum of the "middle" term, number of samples inside the band;
 * loss = sums4[0] / ((C-4) * max(sums4[1], 1e-8)) + sums4[2] / max(sums4[3], 1e-8).  Backward writes
 * g_raw [N,S,C] = g_loss[0] * d loss / d raw (channels 0..3 zero).                                          */
int dmn_penalizer_fwd(const float* raw, const float* z, long z_ray_stride, const float* depth, const float* rays_d,
                      long n_rays, int n_samples, int channels, float tolerance, float deta_w, double* sums4,
                      void* stream);
/* loss [1] from the four sums (fp64 arithmetic, rounded to fp32 at the end): penalizer.py:42-43,52 */
int dmn_penalizer_loss(const double* sums4, int channels, float* loss, void* stream);
int dmn_penalizer_bwd(const float* raw, const float* z, long z_ray_stride, const float* depth, const float* rays_d,
                      long n_rays, int n_samples, int channels, float tolerance, float deta_w, const double* sums4,
                      const float* g_loss, float* g_raw, void* stream);

/* ---- statistics of the instance-matching loss (networks/evaluator.py:19-74) ----------------------------------
 * pred [N, n_cols] (rendered instance map, n_cols = ins_num <= 128), rays grouped by ground-truth class:
 * order [N] (ray indices, class by class), segments [n_classes + 1] (offsets into order), both int64 on the device.
 * stats [3][n_classes][n_cols]: per class and column the sums of log(pred + 1e-8), log(1 - pred + 1e-8), pred,
 * from which the reference's cost_ce / TP / FP / FN follow without its [ins, ins, N] broadcasts.  Backward:
 * classes [N] (class of each ray) and g_stats -> g_pred [N, n_cols].                                         */
int dmn_ins_stats_fwd(const float* pred, const long* order, const long* segments, long n_rays, int n_cols,
                      int n_classes, float* stats, void* stream);
int dmn_ins_stats_bwd(const float* pred, const long* classes, long n_rays, int n_cols, int n_classes,
                      const float* g_stats, float* g_pred, void* stream);

/* The same loss head in five launches and one host round trip (the Hungarian step stays on the host, scipy, as in the
 * reference: evaluator.py:42-52).  max_classes = ins_num (the reference's gt_ins has that many columns, :22).
 *   dmn_ins_label_prep  labels [N] int64 in [0, 1024) -> classes [N] (index of the label among the labels present, in
 *                       increasing order: torch.unique's inverse, :21,25), count [max_classes], meta {V, error flag}
 *   dmn_ins_stats_scan  fp64 partial statistics of dmn_ins_stats_chunks(N) ray chunks from classes [N] (no sorted order
 *                       needed): partial [chunks][3][max_classes][n_cols]
 *   dmn_ins_cost        stats [3][max_classes][n_cols] (the partials summed in chunk order), cost [2][max_classes][n_cols]
 *                       (cost_ce, cost_siou; :60-67), colsum [2][n_cols] (column sums of
 *                       S0 and TP) and host_blob {V, flag, cost_ce + cost_siou [max_classes][n_cols]} for the host
 *   dmn_ins_loss        out4 = {sum, valid_ce, invalid_ce, valid_siou} (:27-37) for col_of [max_classes] / class_of [n_cols]
 *   dmn_ins_loss_bwd    g_stats [3][max_classes][n_cols] from the upstream gradients of the four outputs (NULL = 0)
 * Ray-sharded training (the label set, the statistics and the ray count are the whole batch's; the caller sums them over
 * the shards between the calls): dmn_ins_label_hist (this shard's histogram over labels [0, bins) as floats; `hist`
 * holds bins + 1 floats, the last one is this shard's label-out-of-range flag so that the summed flag is every shard's),
 * dmn_ins_label_slots (classes / counts from the summed histogram [bins + 1]), dmn_ins_stats_sum (partial -> stats, then
 * dmn_ins_cost with partial = NULL), n_rays_dev = the summed ray count on the device (NULL: n_rays).              */
int dmn_ins_label_prep(const long* labels, long n_rays, int max_classes, long* classes, float* count, int* meta,
                       void* stream);
int dmn_ins_label_hist(const long* labels, long n_rays, int bins, float* hist, int* meta, void* stream);
int dmn_ins_label_slots(const long* labels, long n_rays, const float* hist, int bins, int max_classes, long* classes,
                        float* count, int* meta, void* stream);
int dmn_ins_stats_sum(const double* partial, long n_rays, int n_cols, int max_classes, float* stats, void* stream);
int dmn_ins_stats_chunks(long n_rays);
int dmn_ins_stats_scan(const float* pred, const long* classes, long n_rays, int n_cols, int n_classes, double* partial,
                       void* stream);
int dmn_ins_cost(const double* partial, float* stats, const float* count, const int* meta, long n_rays,
                 const float* n_rays_dev, int n_cols, int max_classes, float* cost, float* colsum, float* host_blob,
                 void* stream);
/* The assignment on the device instead of scipy on the host (same algorithm, fp64, scipy's choice among equal
 * candidates): host_blob as dmn_ins_cost wrote it -> match = {col_of [max_classes], class_of [n_cols]} per problem.  Where
 * no assignment exists (labels outside the kernels' table, no finite matching) every entry of that problem's match is -2;
 * dmn_ins_loss then returns NaN terms and dmn_ins_loss_bwd zeros.  meta is read-only (it is shared between calls). */
int dmn_ins_assign(const float* host_blob, const int* meta, int n_cols, int max_classes, int* match, int problems,
                   void* stream);      /* `problems` blobs / match arrays back to back, one workgroup each */
int dmn_ins_loss(const float* cost, const float* colsum, const int* col_of, const int* class_of, const int* meta,
                 long n_rays, const float* n_rays_dev, int n_cols, int max_classes, float* out4, void* stream);
int dmn_ins_loss_bwd(const float* stats, const float* count, const float* colsum, const int* class_of, const int* meta,
                     long n_rays, const float* n_rays_dev, int n_cols, int max_classes, const float* g_total,
                     const float* g_ce, const float* g_inv, const float* g_siou, float* g_stats, void* stream);

/* ---- render_train (networks/render.py:6-28) / manipulator_render (manipulator.py:86-105) --
 * raw [N,S,C], z (stride), rays_d [N,3] -> rgb [N,3], weights [N,S], depth [N],
 * ins [N, C-5] (keep_last = 0, render_train drops the last channel, render.py:26) or
 * ins [N, C-4] (keep_last = 1, manipulator_render keeps it).  Any output pointer may be NULL. */
int dmn_composite_fwd(const float* raw, const float* z, long z_ray_stride, const float* rays_d, long n_rays,
                      int n_samples, int channels, int keep_last, float* rgb, float* weights, float* depth,
                      float* ins, void* stream);

/* Backward of dmn_composite_fwd under the reference's autograd topology (networks/render.py:6-28: the
 * instance map is built from weights.detach(), render.py:22-23): g_raw [N,S,C] from the gradients of
 * rgb_map [N,3], depth_map [N], ins_map [N,K] (needs the forward's `ins` output) and weights [N,S];
 * any gradient pointer may be NULL (= zero).                                                          */
int dmn_composite_bwd(const float* raw, const float* z, long z_ray_stride, const float* rays_d, const float* ins,
                      const float* g_rgb, const float* g_depth, const float* g_ins, const float* g_weights,
                      long n_rays, int n_samples, int channels, int keep_last, float* g_raw, void* stream);

/* ---- sample_pdf (networks/helpers.py:123-155) --------------------------------------------
 * bins [N, n_bins] (row stride bins_ray_stride, 0 = shared), weights [N, n_bins-1] (row stride
 * w_ray_stride, starting at `weights`), u: draw (n,k) = u[n * u_ray_stride + k] (stride 0 = one
 * row shared by all rays, e.g. a torch.linspace(0,1,n_new) table for det=True), or NULL for
 * det=True with the linspace evaluated in the kernel.  out: [N, n_new].                      */
int dmn_sample_pdf(const float* bins, long bins_ray_stride, const float* weights, long w_ray_stride,
                   const float* u, long u_ray_stride, long n_rays, int n_bins, int n_new, float* out, void* stream);

/* ---- hierarchical resampling as called by dm_nerf (networks/render.py:66-70) --------------
 * mids of z_coarse -> sample_pdf(mids, weights[:,1:-1], n_new, u as in dmn_sample_pdf) ->
 * sort(cat([z_coarse, z_new])).  z_fine: [N, n_samples + n_new].  n_new may be 0.            */
int dmn_resample_merge(const float* z_coarse, long z_ray_stride, const float* weights, const float* u,
                       long u_ray_stride, long n_rays, int n_samples, int n_new, float* z_fine, void* stream);

/* ---- torch.sort(torch.cat([a, b], -1)) on depth rows (render.py:70, manipulator.py:190-191) */
int dmn_merge_depths(const float* a, long a_ray_stride, int n_a, const float* b, long b_ray_stride, int n_b,
                     long n_rays, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMNERF_HIP_H_ */
