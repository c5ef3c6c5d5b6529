/* libu2pl_hip.so -- C ABI of the MI355X-native U2PL training hot path.
 *
 * The reference (Haochen-Wang409/U2PL) is pure Python/PyTorch and has NO FFI of
 * its own; its replaceable seams are Python functions.  Each entry point below
 * names the reference code it replaces (file:line under the reference tree).
 * The Python seams that sit on top (u2pl_amd/, same names and argument meaning
 * as the reference) call these through ctypes -- see INTEGRATION.md.
 *
 * Conventions (every function):
 *   - returns 0 on success, a hipError_t value, or U2PL_EINVAL (1001);
 *   - never allocates, frees, synchronises or throws; the caller owns every
 *     buffer (device pointers unless stated) and passes the HIP stream;
 *   - tensors are float32 / int64 / uint32 as in the reference; "NCHW" means
 *     contiguous planar, "rows" means a (pixels, D) view with leading dim ld
 *     (NHWC activations), strides are in ELEMENTS.
 */
#ifndef U2PL_HIP_H
#define U2PL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define U2PL_SEL_MAX_SLOTS 8 /* 2 order statistics per selection spec, <= 4 specs */
/* select workspace words (uint32): [0] n_valid  [1] n_total  [56+j] threshold j (float bits) */
#define U2PL_SEL_WORD_NVALID 0
#define U2PL_SEL_WORD_NTOTAL 1
#define U2PL_SEL_WORD_VAL 40
#define U2PL_SEL_WORD_THR 56

/* ---- reliability.hip ------------------------------------------------------ */
/* F.interpolate(mode="bilinear", align_corners=True): train_semi.py:320-322,345-350,355,372-374 */
int u2pl_bilinear_up_f32(const float* in, long sn, long sc, long sh, long sw, int N, int C, int h, int w,
                         float* out_nchw, int H, int W, hipStream_t stream);
/* autograd of the above for the student branches (train_semi.py:345-355 under loss.backward(), :527) */
int u2pl_bilinear_up_bwd_f32(const float* gout_nchw, int N, int C, int H, int W, float* gin, long sn, long sc,
                             long sh, long sw, int h, int w, hipStream_t stream);
/* F.softmax + torch.max -> (confidence, pseudo label): train_semi.py:323-324 */
int u2pl_pseudo_label_f32(const float* logits_nchw, int N, int C, int H, int W, float* conf, long long* label,
                          hipStream_t stream);
/* entropy = -sum(p*log(p+1e-10)); NaN where label==ignore; *nvalid += #valid:
 * train_semi.py:402-403, loss_helper.py:35-36 (label may be NULL) */
int u2pl_entropy_f32(const float* logits_nchw, const long long* label, int ignore, int N, int C, int H, int W,
                     float* entropy, unsigned* ws, hipStream_t stream);
/* same, fused with the bilinear up-sampling of the LOW-RES logits (train_semi.py:371-374 + 402-403):
 * the full-resolution logit tensor is never materialised; also accumulates ws[0] and the select's hist0 */
int u2pl_entropy_up_f32(const float* in, long sn, long sc, long sh, long sw, int N, int C, int h, int w, int H, int W,
                        const long long* label, int ignore, float* entropy, unsigned* ws, hipStream_t stream);
/* np.percentile(entropy[valid], q) x nspec (exact order statistics + numpy float32 lerp),
 * train_semi.py:405-407,412-415, loss_helper.py:38-40; spec kind 1 = OHEM k-th smallest
 * (loss_helper.py:521-526).  ws must be zeroed, then ws[0]=n_valid, ws[1]=n_total;
 * producers (u2pl_entropy*_f32, u2pl_ohem_prob_f32) also fill the pass-0 histogram (hist0_done=1). */
size_t u2pl_select_workspace_bytes(void);
int u2pl_select_f32(const float* values, long n, int nspec, const int* spec_kind_dev, const float* q32_dev,
                    const long long* kparam_dev, const float* fparam_dev, unsigned* ws, int hist0_done,
                    hipStream_t stream);
/* target[entropy >= thresh] = 255 and count of kept pixels: loss_helper.py:41-44 */
int u2pl_apply_drop_i64(const float* entropy, const unsigned* thr_bits, long long* target, int ignore, long n,
                        unsigned* nkept, hipStream_t stream);
/* low/high entropy masks, nearest down-sampling, label_onehot (batch-slot-0 quirk) as class
 * bitmasks for the concatenated batch: train_semi.py:408-465, utils.py:50-59 */
int u2pl_reliability_masks(const float* entropy, const unsigned* thr_lo_bits, const unsigned* thr_hi_bits,
                           const long long* label_l, const long long* label_u, int ignore, int B, int H, int W,
                           int h, int w, int negative_high_entropy, float* low_mask, float* high_mask,
                           unsigned* lbits, hipStream_t stream);
/* fused tail: unsup target overwrite + masks + class bits in one launch (thr_bits = {drop, low, high}) */
int u2pl_reliability_apply(const float* entropy, const unsigned* thr_bits, const long long* label_l,
                           const long long* label_u, int ignore, int B, int H, int W, int h, int w,
                           int negative_high_entropy, long long* target_u, unsigned* nkept, float* low_mask,
                           float* high_mask, unsigned* lbits, hipStream_t stream);
int u2pl_pack_class_bits(const long long* onehot, int N, int C, int h, int w, unsigned* bits, hipStream_t stream);
int u2pl_unpack_class_bits(const unsigned* bits, int N, int C, int h, int w, long long* onehot,
                           hipStream_t stream);

/* ---- contrast.hip --------------------------------------------------------- */
/* loss_helper.py:103-141 per-pixel masks (anchor / low-valid / negative) as class bitmasks */
int u2pl_contra_classify(const float* prob, long sn, long sc, long sp, const unsigned* lbits,
                         const float* low_mask, const float* high_mask, int N2, int num_labeled, int C, int h,
                         int w, float thr_p, float thr_n, int low_rank, int high_rank, unsigned* abits,
                         unsigned* lowbits, unsigned* nbits, void* compact_workspace, hipStream_t stream);
/* boolean-mask indexing order (loss_helper.py:115-116,119-123,142): idx int32 [3][32][cap] (plane 0 anchors,
   plane 2 negative keys; plane 1 (low-valid) is counted only), counts u32 [3][32].
   compact_workspace (u2pl_compact_workspace_bytes) handed to u2pl_contra_classify receives the per-block
   counts as a by-product; pass counted=1 to u2pl_compact_lists then (NULL / 0 otherwise). */
size_t u2pl_compact_workspace_bytes(long P);
int u2pl_compact_lists(const unsigned* abits, const unsigned* lowbits, const unsigned* nbits, long P, int C,
                       void* workspace, int* idx, long cap, unsigned* counts, int counted, hipStream_t stream);
/* torch.mean(rep_teacher[low_valid], dim=0): loss_helper.py:119-123 */
size_t u2pl_proto_workspace_bytes(long P, int C, int D);
int u2pl_class_prototypes(const float* rows, long ld, int D, const int* idx, long cap, const unsigned* counts,
                          int C, long P, void* workspace, float* proto, const unsigned* lowbits,
                          hipStream_t stream);
/* loss_helper.py:80-154 in THREE launches (classify + per-block counts; prototype streaming; ordered compaction write with
 * in-block offsets + list lengths merged with the ordered prototype finish): same outputs as u2pl_contra_classify +
 * u2pl_compact_lists + u2pl_class_prototypes.  rows: teacher features, row r at rows + r*ld. */
size_t u2pl_contra_phase1_workspace_bytes(long P, int C, int D);
int u2pl_contra_phase1(const float* prob, long sn, long sc, long sp, const unsigned* lbits, const float* low_mask,
                       const float* high_mask, int N2, int num_labeled, int C, int h, int w, float thr_p, float thr_n,
                       int low_rank, int high_rank, const float* rows, long ld, int D, unsigned* abits, unsigned* lowbits,
                       unsigned* nbits, int* idx, long cap, unsigned* counts, float* proto, void* workspace,
                       hipStream_t stream);
/* keys = rep_teacher[negative_mask]: loss_helper.py:142 */
int u2pl_gather_rows_f32(const float* rows, long ld, int D, const int* list, long n, float* out,
                         hipStream_t stream);
/* dequeue_and_enqueue FIFO (utils.py:27-47) on a device ring; tail=(head+len)%cap */
/* The bank as a device-resident object: state = int64 [C][5] {ring offset in `storage` (rows), cap, head, len,
 * ptr}; u2pl_bank_enqueue_f32 appends counts_dev[c] rows per class at the tails and advances the state with the
 * arithmetic of dequeue_and_enqueue (utils.py:27-47) -- no host-side lengths or heads needed.  idx == NULL: class c's
 * rows are rows[row_start_dev[c] + j]. */
size_t u2pl_bank_state_bytes(int C);
int u2pl_bank_init(long long* state, int C, const long long* caps_host, hipStream_t stream);
int u2pl_bank_enqueue_f32(long long* state, float* storage, int D, const float* rows, long ld, const int* idx,
                          long idx_stride, const long long* row_start_dev, const unsigned* counts_dev, int C,
                          hipStream_t stream);
int u2pl_bank_append_f32(float* bank, long cap, long tail, int D, const float* rows, long ld, const int* list,
                         long n_new, hipStream_t stream);
/* same for every class in ONE launch; desc_dev = int64 [nclass][6] {bank, cap, tail, rows, list|0, n_new} */
int u2pl_bank_append_multi_f32(const long long* desc_dev, int nclass, int D, long ld, long max_new,
                               hipStream_t stream);
/* cosine_similarity / temp + cross_entropy(target 0): loss_helper.py:173-230 */
size_t u2pl_infonce_job_bytes(void);
/* head (int [P], all -1 between calls), next (int [njobs*Q]), seg_len (int [njobs*Q], > 0 for the first entry of every
   group of entries of one job that sampled the same candidate; host-built) may all be NULL; when given, the group leaders
   that hit the same pixel are chained for u2pl_scatter_rows_ordered_f32 */
int u2pl_infonce_f32(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K,
                     float temp, float* loss_q, float* ganchor, int* anchor_pix, int* head, int* next,
                     const int* seg_len, hipStream_t stream);
/* u2pl_infonce_f32 + u2pl_infonce_reduce_f32 (+ u2pl_zero_rows_f32 of the rows the previous step's backward wrote: zero_*)
 * in ONE launch.  workspace: u2pl_infonce_fused_workspace_bytes bytes, zeroed once by the caller and reused.  Q % 4 == 0. */
size_t u2pl_infonce_fused_workspace_bytes(int njobs, int Q);
int u2pl_infonce_fused_f32(const void* jobs_dev, int njobs, const float* rep, long ld, int D, int Q, int K, float temp,
                           float* loss_q, float* ganchor, int* anchor_pix, int* head, int* next, const int* seg_len,
                           float* zero_dst, long zero_ld, const int* zero_pix, long zero_n, void* workspace,
                           float inv_valid_seg, float* loss, hipStream_t stream);
int u2pl_infonce_reduce_f32(const float* loss_q, int njobs, int Q, float inv_valid_seg, float* loss,
                            hipStream_t stream);
/* d loss / d rep (loss_helper.py:205-230 backward) without a dense zero fill and without float atomics: dst is a
   persistent all-zero (P, D) buffer; per sampled pixel dst[pix] = scale * gout * (sum of its entries' src rows in
   ascending entry order); head is re-armed to -1.  order / seg_pos / seg_len: host-built grouping of every job's entries
   by sampled candidate (see hipops.group_entries).  u2pl_zero_rows_f32 clears those rows again afterwards. */
int u2pl_scatter_rows_ordered_f32(float* dst, long ld, int D, const int* pix, const int* next, int* head,
                                  const int* order, const int* seg_pos, const int* seg_len, const float* src, long n,
                                  const float* gout_dev, float scale, hipStream_t stream);
int u2pl_zero_rows_f32(float* dst, long ld, int D, const int* pix, long n, hipStream_t stream);

/* ---- losses.hip ----------------------------------------------------------- */
/* F.cross_entropy(ignore_index=255) fwd/bwd: loss_helper.py:46, 295-320, 531 */
size_t u2pl_ce_workspace_bytes(void);
int u2pl_ce_fwd_f32(const float* logits_nchw, const long long* target, int ignore, int N, int C, int H, int W,
                    int unsup_weight, void* workspace, float* out3, hipStream_t stream);
int u2pl_ce_bwd_f32(const float* logits_nchw, const long long* target, int ignore, int N, int C, int H, int W,
                    const float* out3_dev, const float* gout_dev, float gmul, float* grad, hipStream_t stream);
/* nn.CrossEntropyLoss(weight=class_weight, ignore_index, reduction="mean") of the `use_weight: True` criteria
   (loss_helper.py:265-292,461-488): loss = sum w[t]*l / sum w[t]; out3 = {loss, 1/sum w, sum w} */
int u2pl_ce_fwd_weighted_f32(const float* logits_nchw, const long long* target, int ignore, int N, int C, int H, int W,
                             const float* class_weight, void* workspace, float* out3, hipStream_t stream);
int u2pl_ce_bwd_weighted_f32(const float* logits_nchw, const long long* target, int ignore, int N, int C, int H, int W,
                             const float* class_weight, const float* out3_dev, const float* gout_dev, float gmul,
                             float* grad, hipStream_t stream);
/* OhemCrossEntropy2dTensor: loss_helper.py:502-531 */
int u2pl_ohem_prob_f32(const float* logits_nchw, const long long* target, int ignore, int N, int C, int H,
                       int W, float* mask_prob, unsigned* ws, hipStream_t stream);
int u2pl_ohem_apply_i64(const float* mask_prob, const unsigned* thr_bits, const long long* target, int ignore,
                        long n, long long* kept_target, hipStream_t stream);
/* validate(): argmax + intersection/output/target histograms (train_semi.py:620-641, utils.py:568-580) */
int u2pl_confusion_hist_f32(const float* logits_nchw, const long long* target, int ignore, int N, int C, int H, int W,
                            long long* hist3c, hipStream_t stream);

/* ---- conv.hip (implicit GEMM on v_mfma_f32_32x32x2_f32) -------------------- */
/* nn.Conv2d forward (NHWC rows, weights [Cout][R][S][Cin]): resnet.py:25-41,178-186;
 * base.py:23-83; decoder.py:60-106,132-138.  Cin % 32 == 0 (3-channel stem: u2pl_im2col_f32) */
int u2pl_conv2d_fwd_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int N,
                        int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride,
                        int pad, int dil, hipStream_t stream);
/* forward + the train-mode BatchNorm statistics of its output in the conv epilogue (conv -> BN pairs:
 * resnet.py:120-140, base.py:23-83, decoder.py:60-106): per-tile pivot-shifted sums, finished by
 * u2pl_colreduce_finish_f32 */
int u2pl_conv2d_fwd_stat_blocks(int N, int Hout, int Wout, int Cout);
int u2pl_conv2d_fwd_bnstats_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                                int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                int stride, int pad, int dil, const float* pivot, float* stats_partial,
                                hipStream_t stream);
int u2pl_colreduce_finish_f32(const float* partial, int nblk, int C, double* sums, hipStream_t stream);
/* forward + the EVAL-mode BatchNorm (+ residual, ReLU) that follows it, applied in the conv epilogue: the conv -> bn -> relu
 * (-> += identity) chains of resnet.py:118-138, base.py:23-83, decoder.py:60-106 when the model is in eval mode (teacher
 * pseudo-label pass train_semi.py:317-324, validate() :595-654, eval.py).  y = [relu]((conv + bias - mean) * invstd * gamma
 * + beta [+ res]) with the operation order of u2pl_bn_apply_f32, i.e. bit-identical to conv followed by that kernel.
 * res: [N*Hout*Wout][ldr] rows or NULL; Cout % 4 == 0 */
int u2pl_conv2d_fwd_bnact_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int N,
                              int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride,
                              int pad, int dil, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* res, long ldr, int relu, hipStream_t stream);
/* batch independent row-major GEMMs Y_z[M][Nn] = X_z[M][K] * W_z[Nn][K]^T on the fp32 matrix cores (element
   strides zx/zw/zy between the problems): the component products of the Winograd convolutions below */
int u2pl_gemm_batched_f32(const float* x, long ldx, long zx, const float* w, long zw, float* y, long ldy, long zy,
                          long M, int K, int Nn, int batch, hipStream_t stream);
/* Winograd F(mt x mt, 3x3), mt = 2 or 4, for nn.Conv2d(k=3, stride=1, padding=dil, dilation=dil) (resnet.py:25-41,
   base.py:54-83, decoder.py:60-106,132-138): V[a*a][tiles][C] <- x ; U[a*a][O][C] <- w ([O][3][3][C]);
   Mb[a*a][tiles][O] = V . U^T (u2pl_gemm_batched_f32) ; y <- Mb (+bias).  a = mt + 2, tiles = u2pl_wino_tiles.
   transposed = 1 builds U'[a*a][C][O] (taps rotated) for the data gradient.  stats_partial (rows =
   u2pl_wino_stat_blocks) receives the pivot-shifted BatchNorm column sums of y like u2pl_conv2d_fwd_bnstats_f32. */
size_t u2pl_wino_tiles(int N, int H, int W, int dil, int mt);
int u2pl_wino_stat_blocks(long tiles, int O);
int u2pl_wino_input_f32(const float* x, long ldx, int N, int H, int W, int C, int dil, int mt, float* V,
                        hipStream_t stream);
int u2pl_wino_weight_f32(const float* w, int O, int C, int transposed, int mt, float* U, hipStream_t stream);
/* weight-gradient side: Mg[a*a][tiles][O] <- A dY A^T per tile ; batched P_z = Mg_z^T V_z over the tile rows
   (slabs [nsplit][O][a*a][C], nsplit = u2pl_wgrad_batched_splits) ; dw[O][3][3][C] (+)= G^T (sum of slabs) G */
int u2pl_wino_gy_f32(const float* gy, long ldg, int N, int H, int W, int O, int dil, int mt, float* Mg,
                     hipStream_t stream);
int u2pl_wgrad_batched_splits(long M, int Cin, int Cout, int batch);
size_t u2pl_wgrad_batched_workspace_bytes(long M, int Cin, int Cout, int batch);
int u2pl_wgrad_batched_f32(const float* dy, long lddy, long zdy, const float* x, long ldx, long zx, float* part,
                           long M, int Cin, int Cout, int batch, hipStream_t stream);
int u2pl_wino_wgrad_finish_f32(const float* part, int nsplit, int O, int C, int mt, int accumulate, float* dw,
                               hipStream_t stream);
int u2pl_wino_output_f32(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias, float* y,
                         long ldy, float* stats_partial, const float* pivot, hipStream_t stream);
/* Winograd form of u2pl_conv2d_fwd_bnact_f32: the output transform applies the eval-mode BatchNorm (+res, ReLU) */
int u2pl_wino_output_bnact_f32(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias, float* y,
                               long ldy, const float* mean, const float* invstd, const float* gamma, const float* beta,
                               const float* res, long ldr, int relu, hipStream_t stream);
/* BASELINE configs[4] ("config 5": reduced-precision student, fp32 EMA teacher / master weights): the same three
 * convolution products with the operands rounded to bf16 (RNE) while they are staged into LDS and multiplied on the
 * bf16 matrix cores (v_mfma_f32_32x32x16_bf16) with fp32 accumulation; every tensor in HBM stays fp32.  Arguments as the
 * f32 entry points above. */
int u2pl_conv2d_fwd_bf16op_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int N,
                               int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride,
                               int pad, int dil, hipStream_t stream);
int u2pl_conv2d_fwd_bnstats_bf16op_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                                       int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                       int stride, int pad, int dil, const float* pivot, float* stats_partial,
                                       hipStream_t stream);
int u2pl_conv2d_dgrad_bf16op_f32(const float* dy, long lddy, const float* wT, float* dx, long lddx, int N, int Hin,
                                 int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad,
                                 int dil, hipStream_t stream);
int u2pl_conv2d_wgrad_bf16op_f32(const float* dy, long lddy, const float* x, long ldx, float* dw, void* workspace,
                                 int accumulate, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R,
                                 int S, int stride, int pad, int dil, hipStream_t stream);
/* autograd of nn.Conv2d under loss.backward() (train_semi.py:527): data and weight gradients */
int u2pl_conv2d_dgrad_f32(const float* dy, long lddy, const float* wT, float* dx, long lddx, int N, int Hin,
                          int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad,
                          int dil, hipStream_t stream);
int u2pl_weight_transpose_f32(const float* w, float* wt, int Cout, int RS, int Cin, hipStream_t stream);
size_t u2pl_conv2d_wgrad_workspace_bytes(int N, int Hout, int Wout, int Cin, int Cout, int R, int S);
int u2pl_conv2d_wgrad_f32(const float* dy, long lddy, const float* x, long ldx, float* dw, void* workspace,
                          int accumulate, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R,
                          int S, int stride, int pad, int dil, hipStream_t stream);
/* weight-gradient kernel choice under the split arithmetic (A/B switch; env U2PL_WGRAD_TR): 1 (default) = csrc/wgrad_tr.hip
 * (hardware-transposed LDS operand reads, layers with Cout, Cin >= 128), 0 = conv.hip's kernel.  Returns the previous value.
 * The workspace / slab-count queries above follow the switch: query and launch under the same setting. */
int u2pl_wgrad_set_tr(int on);
int u2pl_im2col_f32(const float* x, long ldx, float* col, int Kp, int N, int Hin, int Win, int Cin, int Hout,
                    int Wout, int R, int S, int stride, int pad, int dil, hipStream_t stream);

/* ---- igemm_ws.hip (split-fp32 implicit GEMM with pre-split weights; round 4) -------------------------------------
 * The weights of the network change once per optimizer step (train_semi.py:526-528 optimizer.step(), :531-548 the EMA
 * teacher), the convolutions read them ~14 times per step (resnet.py:120-140 through the two student and two teacher
 * passes and the backward).  u2pl_weight_split3_f32 splits a row-major fp32 matrix [rows][K] (K % 32 == 0) ONCE into the
 * three bf16 piece planes of the split-fp32 arithmetic, stored chunk-major in the image the GEMM's LDS tile has; the
 * *_ws entry points are the conv.hip calls of the same name with that buffer in place of the fp32 weight pointer:
 *   forward:        the [Cout][R*S*Cin] matrix (the weight itself, channels_last)
 *   data gradient:  the transposed [Cin][R*S*Cout] matrix (u2pl_weight_transpose_f32)
 *   Winograd:       the batch of a*a component matrices U[a*a][O][C] of u2pl_wino_weight_f32 (batch = a*a)
 * Same arithmetic, same summation order as conv.hip's split kernels: results are bit-identical to u2pl_conv2d_fwd_f32 /
 * _bnstats / _bnact / u2pl_conv2d_dgrad_f32 / u2pl_gemm_batched_f32 under U2PL_CONV_SPLIT=1 (statistics partial sums
 * included; stats_partial has u2pl_igemm_ws_stat_blocks rows).  Cout > 64 only (narrower layers stay on conv.hip). */
size_t u2pl_weight_split3_bytes(int rows, int K, int batch);
int u2pl_weight_split3_f32(const float* w, long zw, int rows, int K, int batch, void* out, hipStream_t stream);
/* All weight splits / all Winograd filter transforms of a model in ONE launch each (after the optimizer / EMA update; the
 * per-weight entry points above remain the lazy path).  jobs: DEVICE arrays of
 *   struct { const float* src; unsigned short* out; long seg_begin; int rows, Np, K, kind, RS, batch; }   (48 bytes)
 *     kind 0: src = [batch][rows][K]; kind 1: src = conv weight [Cout][RS][Cin] read as [rows = Cin][K = RS * Cout] (the data
 *     gradient's operand); Np = u2pl_weight_split3_pad_rows(rows); seg_begin = prefix sum of batch * Np * K / 8; total = the sum
 *   struct { const float* w; float* U; long begin; int O, C, transposed, mt; }                             (40 bytes)
 *     begin = prefix sum of O * C; total = the sum.
 * Same bits as the per-weight calls. */
int u2pl_weight_split3_multi_f32(const void* jobs, int njobs, long total, hipStream_t stream);
int u2pl_weight_split3_pad_rows(int rows);
int u2pl_wino_weight_multi_f32(const void* jobs, int njobs, long total, hipStream_t stream);
int u2pl_igemm_ws_stat_blocks(int N, int Hout, int Wout);
/* launch shape of the ws kernel (A/B switch, same results; env U2PL_WS_PERSIST): 1 (default) = persistent, at most one
 * block per CU working through its tiles; 0 = one block per tile */
int u2pl_igemm_ws_set_persist(int on);
int u2pl_conv2d_fwd_ws_f32(const float* x, long ldx, const void* wsplit, const float* bias, float* y, long ldy, int N,
                           int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad,
                           int dil, hipStream_t stream);
int u2pl_conv2d_fwd_bnstats_ws_f32(const float* x, long ldx, const void* wsplit, const float* bias, float* y, long ldy,
                                   int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                   int stride, int pad, int dil, const float* pivot, float* stats_partial,
                                   hipStream_t stream);
int u2pl_conv2d_fwd_bnact_ws_f32(const float* x, long ldx, const void* wsplit, const float* bias, float* y, long ldy, int N,
                                 int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride,
                                 int pad, int dil, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* res, long ldr, int relu, hipStream_t stream);
int u2pl_conv2d_dgrad_ws_f32(const float* dy, long lddy, const void* wTsplit, float* dx, long lddx, int N, int Hin,
                             int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad, int dil,
                             hipStream_t stream);
int u2pl_gemm_batched_ws_f32(const float* x, long ldx, long zx, const void* wsplit, float* y, long ldy, long zy, long M,
                             int K, int Nn, int batch, hipStream_t stream);

/* ---- split-fp16 (round 6; csrc/conv_geom.h): the same GEMMs with THREE fp16 piece products per fp32 product instead of six
 * bf16 ones.  Each operand is scaled per tensor by a power of two chosen from its largest magnitude (exact), split into two fp16
 * pieces (x s = h0 + h1 to 2^-25 |x s|) and multiplied as a1 b0 + a0 b1 + a0 b0 with fp32 accumulation; the accumulators are
 * scaled back with ldexp.  Replaces the same reference lines as the *_ws_* family above (u2pl/models/resnet.py:120-140,
 * base.py:54-100, decoder.py:60-142 forward; loss.backward() train_semi.py:527).
 *   "amax object": u2pl_amax_words() (= 2048) device floats, ZEROED by the caller before its producer runs -- 64 shards, one per
 *     128-byte line (same-line atomics serialise; tools/micro/atomic_shard.hip); the tensor's maximum is the maximum over the
 *     shards (bit patterns of |x|: NaN if any element is NaN).  Every x_amax / dy_amax / *_amax argument below is one.
 *   u2pl_absmax_f32: out <- max |x| over [M][C] (C % 4 == 0, ld % 4 == 0); clear != 0 zeroes the object first.  Any upper bound
 *     within ~2^8 of the true maximum keeps fp32-class accuracy; a value BELOW the maximum overflows fp16.
 *   u2pl_conv2d_fwd_bnact_wsh_f32's y_amax: NULL or an amax object for max |y| of the fused output (the next layer's x_amax).
 *   u2pl_weight_split2h_*: planes [K/32][2][Np][32] fp16 + one uint32 per matrix (bit pattern of its max |w|) behind them.
 *     job_scratch: 48 device bytes (the job record of the one-weight call).  The multi call takes the SplitJob table of
 *     u2pl_weight_split3_multi_f32 with out = u2pl_weight_split2h_bytes buffers. */
size_t u2pl_weight_split2h_bytes(int rows, int K, int batch);
int u2pl_weight_split2h_f32(const float* w, long zw, int rows, int K, int batch, void* out, void* job_scratch, hipStream_t stream);
int u2pl_weight_split2h_multi_f32(const void* jobs, int njobs, long total, hipStream_t stream);
int u2pl_amax_words(void);
int u2pl_absmax_f32(const float* x, long ld, long M, int C, float* out, int clear, hipStream_t stream);
/* producers that leave the maximum of what they write (fused: no extra pass): the entry points of the same name without
 * `_amax` plus caller-ZEROED device floats that receive max |output| (bit-pattern atomicMax: deterministic; NULL = not wanted).
 * Reference lines as for the plain forms (BatchNorm base.py:6-8 forward / backward; Winograd transforms of resnet.py:25-41). */
int u2pl_bn_apply_amax_f32(const float* x, long ldx, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* res, long ldr, int relu, const float* drop, long rows_per_image, float* y, long ldy, long M,
                           int C, float* y_amax, hipStream_t stream);
int u2pl_bn_bwd_apply_amax_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy, const float* mean,
                               const float* invstd, const float* gamma, const float* drop, long rows_per_image, const double* sums,
                               double count, float* dx, long lddx, float* dres, long lddr, long M, int C, const double* psums,
                               float* gsink, float* bsink, int accumulate, float* dx_amax, float* dres_amax, const float* relu_beta,
                               hipStream_t stream);
/* y = relu(BN(x)) without a residual: the backward's ReLU mask recomputed from x with the forward's own expression (same bits as
 * [y > 0]) instead of read from y -- u2pl_bn_bwd_sums_f32 / the apply above with y == NULL and relu_beta = the layer's beta */
int u2pl_bn_bwd_sums_mx_f32(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, const float* drop, long rows_per_image, long M, int C,
                            void* workspace, double* sums, hipStream_t stream);
int u2pl_wino_input_amax_f32(const float* x, long ldx, int N, int H, int W, int C, int dil, int mt, float* V, float* v_amax,
                             hipStream_t stream);
int u2pl_wino_output_bnact_amax_f32(const float* Mb, int N, int H, int W, int O, int dil, int mt, const float* bias, float* y,
                                    long ldy, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    const float* res, long ldr, int relu, float* y_amax, hipStream_t stream);
int u2pl_wino_gy_amax_f32(const float* gy, long ldg, int N, int H, int W, int O, int dil, int mt, float* Mg, float* mg_amax,
                          hipStream_t stream);
int u2pl_conv2d_fwd_wsh_f32(const float* x, long ldx, const float* x_amax, const void* wsplit, const float* bias, float* y, long ldy,
                            int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad,
                            int dil, hipStream_t stream);
int u2pl_conv2d_fwd_bnstats_wsh_f32(const float* x, long ldx, const float* x_amax, const void* wsplit, const float* bias, float* y,
                                    long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                    int stride, int pad, int dil, const float* pivot, float* stats_partial, hipStream_t stream);
int u2pl_conv2d_fwd_bnact_wsh_f32(const float* x, long ldx, const float* x_amax, const void* wsplit, const float* bias, float* y,
                                  long ldy, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S,
                                  int stride, int pad, int dil, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, const float* res, long ldr, int relu, float* y_amax, hipStream_t stream);
int u2pl_conv2d_dgrad_wsh_f32(const float* dy, long lddy, const float* dy_amax, const void* wTsplit, float* dx, long lddx, int N,
                              int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int S, int stride, int pad, int dil,
                              hipStream_t stream);
int u2pl_gemm_batched_wsh_f32(const float* x, long ldx, long zx, const float* x_amax, const void* wsplit, float* y, long ldy, long zy,
                              long M, int K, int Nn, int batch, hipStream_t stream);
/* split-fp16 weight gradients (csrc/wgrad_tr.hip; autograd of nn.Conv2d under loss.backward(), train_semi.py:527): the calls
 * u2pl_conv2d_wgrad_f32 / u2pl_wgrad_batched_f32 with the device-scalar maxima of both operands; same workspace and slab plan;
 * only where u2pl_wgrad_h_eligible(Cin, Cout) (Cin, Cout >= 128) -- U2PL_EINVAL otherwise. */
int u2pl_wgrad_h_eligible(int Cin, int Cout);
int u2pl_conv2d_wgrad_h_f32(const float* dy, long lddy, const float* dy_amax, const float* x, long ldx, const float* x_amax, float* dw,
                            void* workspace, int accumulate, int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R,
                            int S, int stride, int pad, int dil, hipStream_t stream);
int u2pl_wgrad_batched_h_f32(const float* dy, long lddy, long zdy, const float* dy_amax, const float* x, long ldx, long zx,
                             const float* x_amax, float* part, long M, int Cin, int Cout, int batch, hipStream_t stream);

/* ---- nn.hip ----------------------------------------------------------------- */
/* nn.SyncBatchNorm / BatchNorm2d (base.py:6-8 get_syncbn): statistics, apply (+residual, ReLU,
 * Dropout2d scale: resnet.py:120-140, decoder.py:79-104), backward; sums are double [2][C] so the
 * cross-rank exchange is one small all-reduce per layer */
size_t u2pl_colreduce_workspace_bytes(long Mseg, int nseg, int C);
int u2pl_bn_stats_f32(const float* x, long ld, long M, int C, const float* pivot, void* workspace, double* sums,
                      hipStream_t stream);
int u2pl_colsum_f32(const float* x, long ld, long Mseg, int nseg, int C, void* workspace, double* sums,
                    hipStream_t stream);
int u2pl_bn_bwd_sums_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                         const float* mean, const float* invstd, const float* drop, long rows_per_image, long M,
                         int C, void* workspace, double* sums, hipStream_t stream);
int u2pl_bn_finalize_f32(const double* sums, double count, const float* pivot, int C, float eps, float momentum,
                         float* mean, float* invstd, float* running_mean, float* running_var, hipStream_t stream);
int u2pl_bn_eval_invstd_f32(const float* running_var, int C, float eps, float* invstd, hipStream_t stream);
/* round 5 (fewer launches on the dependency chain conv -> statistics -> normalise; same arithmetic, same bits):
 *  u2pl_bn_finish_finalize_f32    u2pl_colreduce_finish_f32 + u2pl_bn_finalize_f32 in one launch (single-rank train mode:
 *                                 no all-reduce between them); partial = the conv epilogue's [nblk][2][C] float partial sums;
 *                                 sums_out (double [2C], may be NULL) receives what u2pl_colreduce_finish_f32 would write
 *  u2pl_bn_eval_invstd_multi_f32  u2pl_bn_eval_invstd_f32 for every BatchNorm of a model at once; jobs_dev: device array of
 *                                 u2pl_bn_eval_invstd_job_bytes() = 32-byte records {const float* running_var; float* invstd;
 *                                 int64 first_element; int32 C; float eps}, first_element = running sum of the previous C's,
 *                                 total = sum of C
 *  u2pl_bn_bwd_apply_pg_f32       u2pl_bn_bwd_apply_f32 + the two u2pl_sums_to_f32 calls of the layer's parameter gradients
 *                                 (dgamma = psums[C..2C), dbeta = psums[0..C), psums = the LOCAL backward sums) */
int u2pl_bn_finish_finalize_f32(const float* partial, int nblk, int C, double count, const float* pivot, float eps,
                                float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                double* sums_out, hipStream_t stream);
size_t u2pl_bn_eval_invstd_job_bytes(void);
int u2pl_bn_eval_invstd_multi_f32(const void* jobs_dev, int njobs, long total, hipStream_t stream);
int u2pl_bn_apply_f32(const float* x, long ldx, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, const float* res, long ldr, int relu, const float* drop,
                      long rows_per_image, float* y, long ldy, long M, int C, hipStream_t stream);
int u2pl_bn_bwd_apply_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                          const float* mean, const float* invstd, const float* gamma, const float* drop,
                          long rows_per_image, const double* sums, double count, float* dx, long lddx, float* dres,
                          long lddr, long M, int C, hipStream_t stream);
int u2pl_bn_bwd_apply_pg_f32(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy,
                             const float* mean, const float* invstd, const float* gamma, const float* drop,
                             long rows_per_image, const double* sums, double count, float* dx, long lddx, float* dres,
                             long lddr, long M, int C, const double* psums, float* gsink, float* bsink, int accumulate,
                             hipStream_t stream);
int u2pl_sums_to_f32(const double* sums, int n, float scale, int accumulate, float* out, hipStream_t stream);
/* nn.MaxPool2d(3,2,1,ceil_mode=True): resnet.py:189-191 */
int u2pl_maxpool3s2_fwd_f32(const float* x, long ldx, int N, int H, int W, int C, int Ho, int Wo, float* y, long ldy,
                            unsigned char* tap, hipStream_t stream);
int u2pl_maxpool3s2_bwd_f32(const float* dy, long lddy, const unsigned char* tap, int N, int H, int W, int C, int Ho,
                            int Wo, float* dx, long lddx, hipStream_t stream);
/* torch.cat / slices (base.py:99, decoder.py:117), 1x1 -> HxW bilinear broadcast (base.py:92-94) */
int u2pl_copy_rows_f32(const float* src, long lds, float* dst, long ldd, long M, int C, int accumulate,
                       hipStream_t stream);
int u2pl_copy_cols_f32(const float* src, long lds, float* dst, long ldd, long M, int C, hipStream_t stream);
int u2pl_broadcast_rows_f32(const float* v, long ldv, float scale, float* dst, long ldd, long rows_per_image, long M,
                            int C, hipStream_t stream);
/* F.interpolate(bilinear, align_corners=True) on feature maps: decoder.py:114-116 */
int u2pl_bilinear_rows_fwd_f32(const float* x, long ldx, int N, int h, int w, int C, int H, int W, float* y, long ldy,
                               hipStream_t stream);
int u2pl_bilinear_rows_bwd_f32(const float* dy, long lddy, int N, int h, int w, int C, int H, int W, float* dx,
                               long lddx, hipStream_t stream);
/* F.softmax(pred_all_teacher, dim=1): train_semi.py:365 */
int u2pl_softmax_rows_f32(const float* x, long ldx, float* y, long ldy, long M, int C, hipStream_t stream);
/* torch.optim.SGD step on a flat arena with 3 lr segments (lr_helper.py:18-19, train_semi.py:100-112,528)
 * and the teacher EMA (train_semi.py:531-548) */
int u2pl_sgd_step_f32(float* p, const float* g, float* buf, long n, long b1, long b2, float lr0, float lr1, float lr2,
                      float momentum, float weight_decay, int first, float grad_scale, hipStream_t stream);
int u2pl_ema_update_f32(float* t, const float* s, long n, float decay, float one_minus_decay, hipStream_t stream);
/* torch.optim.Adam step (lr_helper.py:20-21; amsgrad off) on the same arena layout: bias_correction1 = 1 - beta1^t,
 * bias_correction2_sqrt = sqrt(1 - beta2^t) for the step count t kept by the host */
int u2pl_adam_step_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, long b1, long b2, float lr0,
                       float lr1, float lr2, float beta1, float beta2, float eps, float weight_decay,
                       float bias_correction1, float bias_correction2_sqrt, float grad_scale, hipStream_t stream);
/* The whole reliability split in ONE persistent launch (csrc/relfused.hip): bilinear up-sampling + entropy of the
 * train-mode teacher logits (train_semi.py:371-374,402), exact np.percentile thresholds (loss_helper.py:38-40,
 * train_semi.py:405-415), unsup target overwrite (loss_helper.py:41-43), low / high masks + nearest down-sampling +
 * label_onehot class bits (train_semi.py:408-465, utils.py:50-59).  nspec = 1 (target only) or 3; q32_host = HOST
 * array of percentiles/100 in float32.  workspace (u2pl_reliability_fused_workspace_bytes(G) bytes) is zeroed ONCE and
 * reused; epoch = number of earlier launches on it (barrier counters grow monotonically, totals alternate by parity);
 * thresholds are left in workspace words 16..18 (float bits), #kept in word 2.  G = power of two <= #CUs, <= 256.
 * cand: u2pl_reliability_fused_cand_floats(B*H*W, G) floats of scratch.  Every block publishes its entropies sorted by
 * histogram bin, so the members of the bins that hold the ranks are gathered after ONE device-wide barrier; only when
 * those bins hold more values than fit in LDS (all-equal entropies, heavy ties) a second barrier is used (workspace word
 * 4 counts launches, word 5 those that needed it).  flags bit 0: put an agent-scope release / acquire fence pair around
 * the barrier (the published data are written through and read past the L1 already: off by default).
 * Returns 1001 when the shape is not covered (fall back to entropy_up + select + reliability_apply). */
/* kernel launches issued by this library so far (host counter; bench.py: kernel launches per step vs C-ABI calls) */
size_t u2pl_kernel_launches(void);
/* Arithmetic of the fp32 convolution / GEMM entry points: 1 (default; env U2PL_CONV_SPLIT) = every fp32 operand split
 * exactly into three bf16 pieces and the six significant piece products accumulated in fp32 on the bf16 matrix cores
 * (fp32-class accuracy, csrc/conv.hip BF == 3); 0 = v_mfma_f32_32x32x2_f32.  set returns the previous value. */
int u2pl_conv_set_split(int on);
int u2pl_conv_get_split(void);
/* (debug) arm per-block start / end stamps of the three phase-1 kernels: buf = device uint32 [3][4096][2] in 100 MHz
 * ticks (kernel 0 classify, 1 prototype stream, 2 tail); NULL disarms.  Only in a -DU2PL_P1_DBG build of the library
 * (U2PL_EINVAL otherwise); tools/bench_phase1_blocks.py reads it. */
int u2pl_debug_phase1_times(unsigned* buf);
size_t u2pl_reliability_fused_workspace_bytes(int G);
size_t u2pl_reliability_fused_cand_floats(long n_px, int G);
int u2pl_reliability_fused(const float* logits_low, long sn, long sc, long sh, long sw, int B, int C, int h, int w,
                           int H, int W, const long long* label_u, const long long* label_l, int ignore, int nspec,
                           const float* q32_host, int negative_high_entropy, int hm, int wm, float* entropy,
                           long long* target_u, float* low_mask, float* high_mask, unsigned* lbits,
                           unsigned* workspace, float* cand, int G, unsigned epoch, int flags, hipStream_t stream);
/* generate_unsup_data(mode="cutmix"): augmentation.py:498-541 */
int u2pl_cutmix_f32(const float* img, const long long* label, const float* conf, const int* boxes_dev, int B, int C,
                    int H, int W, float* out_img, long long* out_label, float* out_conf, hipStream_t stream);

/* nn.Conv2d(kernel_size=1) on a 1x1 map (ASPP image-pooling branch, base.py:24-28): y[m][o] = x[m][:] . w[o][:] + bias[o]
   for M <= 16 rows, accumulated in float64 (the 2-sample BatchNorm behind it amplifies y's relative error ~50x) */
int u2pl_dense_small_f32(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int M, int K,
                         int Cout, hipStream_t stream);
/* generate_unsup_data(mode="cutout" -> mode 1, boxes) / (mode="classmix" -> mode 2, sel): augmentation.py:486-541.
   sel_dev: uint64 [B], bit c set = class c of image i is in generate_class_mask's selected half */
int u2pl_strong_aug_f32(const float* img, const long long* label, const float* conf, const int* boxes_dev,
                        const unsigned long long* sel_dev, int mode, int B, int C, int H, int W, float* out_img,
                        long long* out_label, float* out_conf, hipStream_t stream);
/* torch.unique(pseudo_labels) (augmentation.py:488) as a per-image presence bitmask; bits [B] zeroed by the caller */
int u2pl_label_presence_i64(const long long* label, int B, long HW, unsigned long long* bits, hipStream_t stream);

/* sliding-window evaluation accumulators (eval.py:184-224): pred [C][H][W] += src [C][hc][wc] at (h0, w0), count += 1;
   then pred /= count */
int u2pl_window_accumulate_f32(float* pred, float* count, int C, int H, int W, const float* src, int h0, int w0,
                               int hc, int wc, hipStream_t stream);
int u2pl_window_normalize_f32(float* pred, const float* count, int C, int H, int W, hipStream_t stream);

/* device-side training data pipeline (augmentation.py:51-266 as composed by cityscapes.py:47-77): ToTensor, Normalize,
   RandResize (bilinear align_corners=False / legacy nearest), flip, zero-padded crop fused into one gather from the
   decoded uint8 sample.  img uint8 [B][H][W][3], lab uint8 [B][H][W], params int32 [B][8] = {rh, rw, flip, pad_top,
   pad_left, crop_y, crop_x, 0} (device); mean3 / std3 are HOST pointers to three floats. */
int u2pl_augment_u8_f32(const unsigned char* img, const unsigned char* lab, const int* params, int B, int H, int W,
                        int Sh, int Sw, const float* mean3, const float* std3, float* out_img, long long* out_lab,
                        hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* U2PL_HIP_H */
