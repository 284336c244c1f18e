/*
 * tocvp.h -- C-ABI of libtocvp.so: hand-written HIP (gfx950 / MI355X) kernels for the TextOCVP
 * autoregressive slot-rollout hot path (SAVi encode -> slot attention -> text-conditioned
 * predictor -> spatial-broadcast decoder).
 *
 * The reference (angelvillar96/TextOCVP) has NO native layer: every op on the path is a stock
 * ATen call issued from Python.  This header therefore does not
 * replace an existing FFI; it is the FFI a maintainer would bind from the reference's
 * src/models modules (ctypes stub in INTEGRATION.md).  Each entry point names the reference
 * operator(s) it replaces as `path:line` relative to /root/reference/src.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers + sizes; no torch / STL types.
 *   - every pointer is a DEVICE pointer owned by the caller (fp32 unless stated), row-major,
 *     feature-last: (B,N,D), (B,K,D), NHWC for conv activations.
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued: no allocation,
 *     no synchronisation, no global state -> re-entrant per stream and graph-capturable.
 *   - return 0 on success, a negative TOCVP_E* code otherwise (nothing is enqueued on error).
 */
#ifndef TOCVP_H
#define TOCVP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOCVP_VERSION 100 /* major*10000 + minor*100 + patch */

#define TOCVP_OK 0
#define TOCVP_EINVAL (-1)   /* bad argument (null pointer, size, unsupported shape) */
#define TOCVP_ELAUNCH (-2)  /* hipLaunch reported an error */
#define TOCVP_EALIGN (-3)   /* pointer / leading dimension not aligned as required */

/* epilogue activation of tocvp_gemm_f32 */
#define TOCVP_ACT_NONE 0
#define TOCVP_ACT_RELU 1
#define TOCVP_ACT_GELU 2 /* exact erf GELU (nn.TransformerEncoderLayer activation="gelu") */
#define TOCVP_ACT_GATE 3 /* tocvp_gemm_bf16wfrag_f32 only: y = (R > 0) ? x W^T + b : 0 -- R is read as the gate, not
                          * added: the data gradient taken through the ReLU whose output is R (training step) */

int tocvp_version(void);
/* human-readable text for a TOCVP_E* code */
const char* tocvp_strerror(int code);

/* ---------------------------------------------------------------------------------------------
 * C[M,N] = act(A[M,K] * W[N,K]^T + bias[N] + rowvec[idx(row)][N]) + R[M,N]       (fp32 MFMA)
 *
 * Replaces every nn.Linear on the path: encoder_mlp (models/SAVi.py:115-120), to_q/to_k/to_v,
 * GRU input/hidden projections and slot MLP (models/Blocks/attention.py:54-64,105-110),
 * q/k/v/out projections and MLPs of TransformerBlock / AdaptedEncoderBlock
 * (attention.py:167-175,296-300,355-359,428-432), mlp_in / mlp_out
 * (models/Predictors/text_cond_OCVP.py:49-50), and the text encoder's linears
 * (models/EncodersDecoders/text_encoders.py:44-69).
 *
 *   W is in nn.Linear layout (N rows of K).  lda/ldr/ldc are row strides in floats.
 *   bias, R, rowvec may be NULL.  rowvec is (rv_mod, N): idx(row) = (row / rv_div) % rv_mod,
 *   reversed (rv_mod-1-idx) when rv_flip != 0 -- this is the flipped learned temporal
 *   positional encoding of models/Blocks/model_blocks.py:358-379.
 *   Requirements: K % 4 == 0, lda % 4 == 0, A and W 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int tocvp_gemm_f32(const float* A, int lda, const float* W, const float* bias,
                   const float* R, int ldr,
                   const float* rowvec, int rv_div, int rv_mod, int rv_flip,
                   float* C, int ldc, int M, int N, int K, int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Same contract as tocvp_gemm_f32 (fp32 A, C, bias, R, rowvec in HBM) with SPLIT-bf16 operands on
 * the bf16 matrix cores:
 *   nsplit = 2 ("bf16x3"): x = hi + lo,       hh + hl + lh            (~2^-16 per product)
 *   nsplit = 3 ("bf16x6"): x = hi + mid + lo, hh + hm + mh + hl + mm + lh   (fp32-class accuracy)
 * A is split while staged into LDS; Wsplit = (N, nsplit, K) bf16 planes from
 * tocvp_split_weights_bf16 (done once per weight).  Requires K % 32 == 0.
 * ------------------------------------------------------------------------------------------- */
int tocvp_split_weights_bf16(const float* w, void* out, int N, int K, int nsplit, void* stream);
int tocvp_gemm_bf16split_f32(const float* A, int lda, const void* Wsplit, int nsplit,
                             const float* bias, const float* R, int ldr,
                             const float* rowvec, int rv_div, int rv_mod, int rv_flip,
                             float* C, int ldc, int M, int N, int K, int act, void* stream);

/* Variant with the weight stored in MFMA-fragment order Wf[n/32][k/16][plane][lane] (16 B per lane,
 * tocvp_split_weights_frag_bf16): B fragments are fetched straight from L2 with coalesced 1 KiB
 * loads and never pass through LDS.  Requires K % 64 == 0 and N % 32 == 0.  With a_split (A given as operand
 * planes (M, planes, K)) one call takes at most 2^32 bytes of planes (32-bit byte offsets; TOCVP_EINVAL beyond: cut
 * the rows).  The SAME limit holds for fp32 A in every split GEMM of this family (tocvp_gemm_bf16split_f32,
 * tocvp_gemm_bf16wfrag_f32 with a_split == 0, tocvp_gemm_f16wfrag_f32, tocvp_gemm_f16wfrag_ws_f32):
 * M * lda * 4 < 2^32, TOCVP_EINVAL beyond (rows behind the limit came back wrong, silently, until round 4; the
 * exact-fp32 tocvp_gemm_f32 has no such limit). */
int tocvp_split_weights_frag_bf16(const float* w, void* out, int N, int K, int nsplit, void* stream);
int tocvp_gemm_bf16wfrag_f32(const void* A, int a_split, int lda, const void* Wfrag, int nsplit,
                             const float* bias, const float* R, int ldr,
                             const float* rowvec, int rv_div, int rv_mod, int rv_flip,
                             void* C, int c_split, int ldc, int M, int N, int K, int act,
                             void* stream);
/* "f16x3": the same kernel with fp16 planes (x = hi + lo in fp16, 22 significant bits, products
 * hh + hl + lh): fp32-class accuracy at HALF the MFMA count of bf16x6, valid while |x| < 65504. */
int tocvp_split_weights_frag_f16(const float* w, void* out, int N, int K, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused predictor MLP (round 4, csrc/mlp_fused.hip):  Y = relu(X W1^T + b1) W2^T + b2 (+ R), the nn.Linear -> ReLU ->
 * nn.Linear pairs of the reference's predictor blocks (models/Blocks/attention.py:355-359 applied at :395 / :521-523,
 * and :428-432 applied at :461-463), in the f16x3 arithmetic of tocvp_gemm_bf16wfrag_f32 and bit-identical to that
 * entry point called twice (c_split = 1 hidden planes in between): a workgroup owns 128 rows and walks the hidden
 * dimension in chunks of 128, so the (M, Hd) hidden activation never reaches HBM.
 *   x_planes: (M, 2, E) fp16 operand planes of 2^8 x (tocvp_layernorm_split_bf16 with split = 22, or any c_split
 *   producer); w1_frag / w2_frag: tocvp_split_weights_frag_f16 of W1 (Hd, E) / W2 (E, Hd); b1 (Hd), b2 (E); R (M, E)
 *   row stride ldr, or NULL; Y (M, E) row stride ldy.  E == 512, Hd % 128 == 0, M * E * 4 < 2^32.
 *   Valid for |x|, |hidden| < 255, |w| < 63 (saturating beyond, like every fp16-plane kernel).
 *   ws (nullable): tocvp_mlp_f16x3_fused_ws_bytes() bytes, zero on first use, one per stream.  With it the tiles of a
 *   last, partly filled round of workgroups are cut into slices of the hidden dimension whose partial sums meet in the
 *   workspace (added in slice order by the last arriver: deterministic; those rows then differ from the uncut sum in
 *   the last bits, as any other summation order does).  Without it every tile is one workgroup.
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_mlp_f16x3_fused_ws_bytes(void);
int tocvp_mlp_f16x3_fused_f32(const void* x_planes, const void* w1_frag, const float* b1, const void* w2_frag,
                              const float* b2, const float* R, int ldr, float* Y, int ldy, int M, int E, int Hd,
                              void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * f16x3 GEMM with the activation chunk resident in LDS (round 4, csrc/gemm_f16c.hip): C = act(A W^T + bias) (+ R) for
 * wide products (N % 512 == 0 or N % 384 == 0, K % 128 == 0) -- nn.Linear 1024 -> 1024 of the reference's MLPPatchDecoder
 * (models/EncodersDecoders/decoders.py:264-307) and the wide projections of the predictor / ViT blocks
 * (models/Blocks/attention.py:167-175).  A workgroup owns 128 rows x 512 outputs: the A operand is walked in 128-deep
 * chunks (64 KB images, double buffered, LDS-DMA), the weights stream from L2 in fragment order straight into the MFMA
 * operand registers.  Bit-identical to tocvp_gemm_bf16wfrag_f32 (nsplit 22, a_split = 1) on the same operands.
 *   A_planes: (M, 2, K) fp16 planes of 2^8 a (any c_split producer); W_frag: tocvp_split_weights_frag_f16 of W (N, K);
 *   bias (N) or NULL; R (M, N) row stride ldr or NULL (added after the activation); C: fp32 (M, N) row stride ldc, or
 *   (c_split) fp16 planes (M, 2, N); act: TOCVP_ACT_NONE / RELU / GELU.  M * K * 4 < 2^32.
 * ------------------------------------------------------------------------------------------- */
int tocvp_gemm_f16chunk_f32(const void* A_planes, const void* W_frag, const float* bias, const float* R, int ldr,
                            void* C, int c_split, int ldc, int M, int N, int K, int act, void* stream);
/* The same structure for mid-size row counts (600 .. 10 000 rows: the predictor's products at small evaluation batches,
 * models/Blocks/attention.py:167-175, 355-359, 428-432): 64 rows x 256 outputs per workgroup, two workgroups per CU, A in
 * 128-deep chunks by LDS-DMA, weights from L2 in fragment order; N % 256 == 0, K % 128 == 0.  Bit-identical to
 * tocvp_gemm_bf16wfrag_f32 on the same operands.  Arguments as tocvp_gemm_f16chunk_f32.  (A split-K form through a
 * workspace existed in round 4, measured slower than the unsplit launch and was retired.) */
int tocvp_gemm_f16mid_f32(const void* A_planes, const void* W_frag, const float* bias, const float* R, int ldr, void* C,
                          int c_split, int ldc, int M, int N, int K, int act, void* stream);
int tocvp_gemm_f16wfrag_f32(const void* A, int lda, const void* Wfrag, const float* bias,
                            const float* R, int ldr, const float* rowvec, int rv_div, int rv_mod,
                            int rv_flip, void* C, int ldc, int M, int N, int K, int act, void* stream);
/* Same product with a caller-owned workspace (tocvp_gemm_wfrag_ws_bytes() bytes, 16-byte aligned, ZERO before its
 * first use, one per stream that may run concurrently): with few output tiles (small evaluation batches: M <= ~600
 * rows) the kernel splits K over idle CUs -- up to 16 slices, each workgroup parks its raw accumulators in `ws` and
 * the last arriver of a tile adds the slices in slice order (deterministic; arrival counters re-arm themselves) and
 * runs the epilogue.  ws == NULL: never split.  A is fp32 (M, K) with row stride lda; c_split as below. */
size_t tocvp_gemm_wfrag_ws_bytes(void);
int tocvp_gemm_f16wfrag_ws_f32(const void* A, int lda, const void* Wfrag, const float* bias, const float* R,
                               int ldr, const float* rowvec, int rv_div, int rv_mod, int rv_flip, void* C,
                               int c_split, int ldc, int M, int N, int K, int act, void* ws, size_t ws_bytes,
                               void* stream);
/*   a_split != 0: A is already split by its producer, (M, nsplit, K) bf16 planes (lda ignored);
 *   c_split != 0: C is written as (M, nsplit, N) bf16 planes for a following split GEMM
 *   (ldc ignored).  Splitting an activation once in its producer instead of once per column block
 *   of every consumer removes the VALU work that bounded the in-kernel split. */

/* ---------------------------------------------------------------------------------------------
 * Range check of the fp16-plane arithmetic (the one checked pass after weights are loaded): `out` (one 32-bit word of
 * device memory) receives the BIT PATTERN of max |x[i]| over n floats, NaN counted as +inf.  Replaces the
 * torch abs / max reductions the check used to run.
 * ------------------------------------------------------------------------------------------- */
int tocvp_absmax_f32(const float* x, long n, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * y[r,:] = LayerNorm(x[r,:] + add[r % add_rows,:]) * gamma + beta      (biased variance, eps)
 * Replaces nn.LayerNorm at attention.py:49-51,361-362,427,435-436, SAVi.py:116 and
 * text_encoders.py:63,65-68; `add` (may be NULL) fuses SoftPositionEmbed's addend
 * (model_blocks.py:215-226).  D <= 1024, D % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int tocvp_layernorm_f32(const float* x, const float* add, int add_rows, const float* gamma,
                        const float* beta, float* y, int rows, int D, float eps, void* stream);
/* same, output written as operand planes of a split GEMM: nsplit 2 or 3 -> (rows, nsplit, D) bf16;
 * nsplit 22 -> (rows, 2, D) fp16 planes of 2^8 y (the f16x3 arithmetic of tocvp_gemm_f16wfrag_f32) */
int tocvp_layernorm_split_bf16(const float* x, const float* add, int add_rows, const float* gamma,
                               const float* beta, void* ysplit, int nsplit, int rows, int D,
                               float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention  O = softmax(Q K^T * scale [+ key-padding mask]) V        (fp32 MFMA,
 * flash-style online softmax).  Replaces MetaAttention.attention/split_into_heads/merge_heads
 * (attention.py:183-215) for MultiHeadSelfAttention (:245-265), MultiHeadCrossAttention
 * (:303-319) and nn.MultiheadAttention inside nn.TransformerEncoderLayer (text_encoders.py:44-51).
 *
 *   Q:(B,Tq,H*dh) row stride ldq; K,V:(B,Tk,H*dh) row strides ldk, ldv; O row stride ldo.
 *   Batch strides are Tq*ldq, Tk*ldk, Tk*ldv, Tq*ldo.  dh in {32, 64}.
 *   key_len (int32, B entries, may be NULL): keys j >= key_len[b] are excluded
 *   (src_key_padding_mask of text_encoders.py:104-105).
 * ------------------------------------------------------------------------------------------- */
int tocvp_mha_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                  float* O, int ldo, int B, int H, int Tq, int Tk, int dh, float scale,
                  const int32_t* key_len, void* stream);
/* same as tocvp_mha_f32 with both products (Q K^T and P V) on the f16 matrix cores (f16x3 split operands,
 * fp32-class, |q|, |k|, |v| < 255); the softmax stays fp32 */
int tocvp_mha_qk16_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                       float* O, int ldo, int B, int H, int Tq, int Tk, int dh, float scale,
                       const int32_t* key_len, void* stream);
/* Sequence lengths one past a multiple of the 128-query tile (the ViT's 256 patches + class token, timm_encoders.py:59-70):
 * tocvp_mha_qk16_rows_f32 = tocvp_mha_qk16_f32 on the FIRST q_rows query rows of tensors that hold Tq_total rows per sample
 * (batch strides Tq_total * ldq / ldo), tocvp_mha_one_query_f32 = row q_row of every sample against all keys in exact fp32
 * on the vector ALUs (one wave per (sample, head); Tk <= 1024).  Together they replace a third query tile that staged all
 * keys and values again for one row. */
int tocvp_mha_qk16_rows_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                            int B, int H, int Tq_total, int q_rows, int Tk, int dh, float scale, const int32_t* key_len,
                            void* stream);
int tocvp_mha_one_query_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                            int B, int H, int Tq_total, int q_row, int Tk, int dh, float scale, const int32_t* key_len,
                            void* stream);
/* the two entries above with O written as fp16 operand planes (B * Tq_total, 2, H*dh) of 2^8 O for a following plane-input
 * GEMM (the ViT's attention output projection, timm_encoders.py:59-70): both products stay on the f16 matrix cores */
int tocvp_mha_qk16_rows_split_f16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* Osplit,
                                  int B, int H, int Tq_total, int q_rows, int Tk, int dh, float scale,
                                  const int32_t* key_len, void* stream);
int tocvp_mha_one_query_split_f16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* Osplit,
                                  int B, int H, int Tq_total, int q_row, int Tk, int dh, float scale, const int32_t* key_len,
                                  void* stream);
/* Round 5 -- the same attention with q / k / v given as fp16 OPERAND PLANES, as the epilogue of the projection GEMM writes
 * them (tocvp_gemm_* with c_split: rows [hi of all columns | lo of all columns] of 2^8 x): MultiHeadSelfAttention.forward
 * (attention.py:245-265) in the predictor blocks, the ViT blocks of timm_encoders.py:59-70.  dh = 64 only.
 *   Q / K / V: f16 pointers ALREADY offset to the first column of the q / k / v block; ldq / ldk / ldv = plane stride in
 *   elements (= the projection's output width; a token row is 2 ld* elements; multiples of 8); samples are Tq_total (Q, O) and
 *   Tk (K, V) rows apart.  Exactly one of O (fp32 rows, stride ldo) and Osplit (fp16 planes (B * Tq_total, 2, H * 64)) is given.
 *   The launch covers the first q_rows query rows of every sample.  (Sequence lengths of 128 n + 1 simply take a third,
 *   nearly empty query block here: its keys / values are L2 hits and the kernel no longer splits or transposes them, 346 us
 *   at 256 x 12 x 257 against 516 us with a one-row vector-ALU kernel beside it.)
 * The kernel only copies the planes into LDS (no split, no transpose: V fragments by ds_read_b64_tr_b16) and is otherwise
 * the arithmetic of tocvp_mha_qk16_f32 expression for expression: BIT-IDENTICAL to it on the values the planes encode. */
int tocvp_mha_planes_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, float* O, int ldo,
                         void* Osplit, int B, int H, int Tq_total, int q_rows, int Tk, int dh, float scale,
                         const int32_t* key_len, void* stream);
/* same, O written as operand planes: (B*Tq, nsplit, H*dh) bf16 (nsplit 2 or 3) or, nsplit 22,
 * (B*Tq, 2, H*dh) fp16 planes of 2^8 O */
int tocvp_mha_split_bf16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                         void* Osplit, int nsplit, int B, int H, int Tq, int Tk, int dh, float scale,
                         const int32_t* key_len, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One slot-attention iteration over the feature grid (the north-star kernel), attention.py:99-103:
 *     dots = q k^T * scale; attn = softmax_over_slots(dots) + eps;
 *     updates = (attn / sum_N attn) v
 * q:(B,Ks,D) contiguous; k,v:(B,N,D) with row stride ldkv (k and v may be the two halves of one
 * fused (B,N,2D) projection); updates:(B,Ks,D).  attn_out (may be NULL): (B,Ks,N) = attn after
 * "+eps" (SlotAttention.attention_masks side effect, attention.py:101).
 * Ks <= 32, D == 128, N % 32 == 0.  ws: workspace of tocvp_slot_attn_ws_bytes(B,N) bytes, 16-byte aligned
 * (ticket words + one 16.5 KB partial record per workgroup).  The ticket words must be ZERO on entry
 * (tocvp_slot_attn_ws_init once after allocation, or whenever B changes) and are zero again on return: the
 * last-arriving workgroup of a sample resets its word, so iterations need no memset between them.
 * ONE launch: k and v are streamed exactly once with coalesced
 * 16-byte loads, both contractions run on the f16 matrix cores with split fp16 operands (fp32-class,
 * valid for |q * scale|, |k|, |v| < 255), the last-arriving workgroup of a sample adds the partial
 * records in a fixed order (deterministic) and renormalises.
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_slot_attn_ws_bytes(int B, int N);
int tocvp_slot_attn_ws_init(void* ws, size_t ws_bytes, void* stream);
int tocvp_slot_attn_iter_f32(const float* q, const float* k, const float* v, int ldkv,
                             float* updates, float* attn_out, int B, int Ks, int N, int D,
                             float scale, float eps, void* ws, size_t ws_bytes, void* stream);
/* Same iteration with k / v already in fp16 operand planes, as the fused [to_k; to_v] projection writes them
 * through tocvp_gemm_bf16wfrag_f32(c_split = 1, f16x3): kv_planes (B, N, 2 planes, 2 D) fp16 of 2^8 * value,
 * i.e. one 1 KiB row [k hi | v hi | k lo | v lo] per location.  Same HBM bytes as fp32 k / v, no operand
 * conversion inside the streaming loop (HBM-bound instead of issue-bound). */
int tocvp_slot_attn_iter_planes_f32(const float* q, const void* kv_planes, float* updates, float* attn_out,
                                    int B, int Ks, int N, int D, float scale, float eps, void* ws,
                                    size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * nn.GRUCell gate math (attention.py:105-108), gate order (r,z,n):
 *   gi = W_ih x + b_ih, gh = W_hh h + b_hh  (both (rows,3D), from tocvp_gemm_f32)
 *   out = (1-z)*n + z*h
 * ------------------------------------------------------------------------------------------- */
int tocvp_gru_gates_f32(const float* gi, const float* gh, const float* h, float* out, int rows,
                        int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SoftPositionEmbed addend (model_blocks.py:186-226, grid of model_utils.py:12-34):
 *   pos[y,x,c] = sum_j grid[y,x,j] * w[c,j] + b[c],  grid = [gy, gx, 1-gy, 1-gx], g=linspace(-1,1)
 * w:(C,4) (the 1x1 conv weight), out:(H,W,C).
 * ------------------------------------------------------------------------------------------- */
int tocvp_pos_embed_f32(const float* w, const float* b, float* out, int H, int W, int C,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * First encoder layer: Conv2d(3->Cout, k5, p2) + ReLU reading the video tensor in place
 * (encoders.py:130-154, layer 0).  x: image n at x + n*img_stride floats, planes (C,H,W)
 * contiguous; w: nn.Conv2d layout (Cout,3,5,5); y: NHWC (nimg,H,W,Cout).  Cout == 32.
 * ------------------------------------------------------------------------------------------- */
int tocvp_conv5x5_in3_f32(const float* x, long long img_stride, const float* w, const float* bias,
                          float* y, int nimg, int H, int W, int Cout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conv2d(Cin->Cout, k5, p2) [+ReLU] as an implicit GEMM on fp32 MFMA, NHWC in / NHWC out.
 * Replaces ConvBlock (model_blocks.py:49-108) in SimpleConvEncoder (encoders.py:130-154) and
 * ConvDecoder (decoders.py:96-110).
 *   wp: weights repacked as (25 taps, Cout, Cin) (tocvp_pack_conv_weights_f32).
 *   Cin in {32,64,128}, Cout in {32,64}, H % 8 == 0, W % 32 == 0.
 * Collapsed-input mode (in_mode = 1) implements the decoder's layer 0 analytically and feeds
 * layer 1 without ever materialising SAVi.broadcast's (B*K,D,H,W) tensor (SAVi.py:264-275):
 *   x_in[n,y,x,c] = relu(cpos[y,x,c] + S[n, cls(y,x), c]),
 *   cpos = conv0(pos_embed) + bias0 (slot independent), S[n,cls,:] = tapsum[cls] * slot_n,
 *   cls(y,x) in 5x5 border classes (which of the 25 taps fall inside the image).
 *   In that mode x = cpos (H,W,Cin), aux = S (nimg,25,Cin).
 * ------------------------------------------------------------------------------------------- */
int tocvp_pack_conv_weights_f32(const float* w, float* wp, int Cout, int Cin, int ksize,
                                void* stream);
int tocvp_conv5x5_f32(const float* x, const float* aux, int in_mode, const float* wp,
                      const float* bias, float* y, int nimg, int H, int W, int Cin, int Cout,
                      int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Same convolution (Cin == Cout == 64 only, fp32 tensors in HBM on both sides) evaluated with
 * SPLIT-bf16 operands on the bf16 matrix cores: x = hi + lo (two bf16), products as
 * hi*hi + hi*lo + lo*hi with fp32 accumulation ("bf16x3", per-product relative error ~2^-16).
 * 5.3x fewer matrix-pipe cycles per FLOP than the fp32 MFMA path of tocvp_conv5x5_f32.
 *   wsplit: weights pre-split by tocvp_split_conv_weights_bf16 as (25, Cout, [hi Cin | lo Cin]) bf16.
 *   in_mode / aux as in tocvp_conv5x5_f32.
 * ------------------------------------------------------------------------------------------- */
int tocvp_split_conv_weights_bf16(const float* w, void* out, int Cout, int Cin, void* stream);
/* same weights in MFMA-fragment order (25, 2 passes, 2 k-steps, 2 column blocks, 2 planes, 64 lanes, 8):
 * the kernel then fetches B fragments straight from L1/L2 (no weight image in LDS, no barrier in
 * the tap loop).  Pass either layout (or both; wfrag is preferred).
 * relu: 0 none, 1 ReLU, 2 = gate: in_mode 0 with wfrag only, y is zeroed where aux (nimg,H,W,Cout) <= 0 --
 * the data gradient of the training step's frozen decoder taken through the ReLU of the layer below
 * (aux = that layer's activation) without a separate masking pass. */
int tocvp_split_conv_weights_frag_bf16(const float* w, void* out, int Cout, int Cin, void* stream);
int tocvp_conv5x5_bf16x3_f32(const float* x, const float* aux, int in_mode, const void* wsplit,
                             const void* wfrag, const float* bias, float* y, int nimg, int H, int W,
                             int Cin, int Cout, int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Same convolution with HYBRID f16 + fp8 split operands ("f16f8", conv_f16f8.hip): with X = 2^8 x,
 * W = 2^10 w, Xh = f16(X), Wh = f16(W) the product is evaluated as Xh*Wh on the f16 matrix cores plus
 * the two cross terms X*(W-Wh) + (X-Xh)*W on the fp8 (OCP e4m3, block-scaled 32x32x64) matrix cores
 * at twice the f16 rate: 2/3 of the matrix cycles of bf16x3 for ~2.3x its (2^-16 class) error.
 * Valid for |x| < 255 and |w| < 63 (operands saturate beyond).  in_mode / aux as tocvp_conv5x5_f32.
 *   wf16 / wf8: fragment-order weight images written by tocvp_split_conv_weights_f16f8, of
 *   tocvp_conv_weights_f16f8_bytes(0) / (1) bytes.
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_conv_weights_f16f8_bytes(int which);
int tocvp_split_conv_weights_f16f8(const float* w, void* wf16, void* wf8, int Cout, int Cin,
                                   void* stream);
/* layout: bit 0 = x, bit 1 = y in the pass-major activation layout (n, 4, H, W, 16) (channel c of a
 * pixel at plane c / 16) instead of NHWC.  The kernel consumes 16 channels per pass: NHWC makes every
 * pass touch one 64-byte quarter of each pixel and whole 128-byte lines move twice; pass-major keeps
 * the reads of a pass contiguous.  Private format between consecutive decoder layers. */
int tocvp_conv5x5_f16f8_f32(const float* x, const float* aux, int in_mode, const void* wf16,
                            const void* wf8, const float* bias, float* y, int nimg, int H, int W,
                            int Cin, int Cout, int relu, int layout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Same convolution with SPLIT-fp16 operands ("f16x3", conv_f16x3.hip), the default decoder arithmetic:
 * X = 2^8 x = Xh + Xl, W = 2^10 w = Wh + Wl in fp16 planes, product = Xl*Wh + Xh*Wl + Xh*Wh on the f16
 * matrix cores (three v_mfma_f32_32x32x16_f16 into one fp32 accumulator): ~2^-21 per product, i.e.
 * fp32-class.  Replaces nn.Conv2d(64,64,5,padding=2)+ReLU of ConvDecoder (decoders.py:96-110).
 * Valid for |x| < 255 and |w| < 63 (operands saturate beyond).  in_mode / aux as tocvp_conv5x5_f32,
 * layout bits 0 / 1 as tocvp_conv5x5_f16f8_f32; bit 2 is refused (TOCVP_EINVAL: the persistent form of rounds 2-4 is retired);
 * bit 3 (with bit 0 and / or 1): the pass-major buffers hold fp16 OPERAND PLANES -- per pixel and
 * 16-channel pass the 64 bytes [16 f16 Xh | 16 f16 Xl] of 2^8 x instead of 16 floats.  A layer called with bits 1 + 3
 * writes them from its epilogue (the split the next layer's staging would make of the same values), a layer called
 * with bits 0 + 3 stages them into LDS by DMA without converting: the chain gives bit-identical results to the fp32
 * hand-over.  Such a buffer is only meaningful as the input of this entry point.
 *   wf: fragment-order weight image written by tocvp_split_conv_weights_dec_f16x3, of
 *   tocvp_conv_weights_dec_f16x3_bytes() bytes.
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_conv_weights_dec_f16x3_bytes(void);
int tocvp_split_conv_weights_dec_f16x3(const float* w, void* wf, int Cout, int Cin, void* stream);
int tocvp_conv5x5_dec_f16x3_f32(const float* x, const float* aux, int in_mode, const void* wf,
                                const float* bias, float* y, int nimg, int H, int W, int Cin, int Cout,
                                int relu, int layout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same decoder conv (nn.Conv2d(64, 64, 5, padding=2) + ReLU, models/EncodersDecoders/decoders.py:96-110) as a
 * vertical Winograd F(4, 5) with five direct horizontal taps (round 5, csrc/conv_wino.hip): 40 split-fp16 matrix
 * products per four output pixels instead of 100, transforms in fp32.  Not bit-identical to the entry above (another
 * summation); scripts/probes/winograd_numerics.py: same error class against fp64 (4.1e-7 vs 3.1e-7 of the activation
 * maximum after three layers).
 *   Weights: tocvp_split_conv_weights_wino_f16x3 is called twice -- scales == NULL: absmax_out[8] (device) receives
 *   max |U_xi| of the eight transformed weight rows; then the HOST picks powers of two scales[xi] with
 *   scales[xi] * max |U_xi| <= 2^14 and calls again with them (host array) to get wf
 *   (tocvp_conv_weights_wino_f16x3_bytes() bytes, fragment order).
 *   coef (host array of 32 floats): AT[a][xi] / (16 * scales[xi]), AT = the 4 x 8 output transform over the points
 *   0, 1, -1, 2, -2, 1/2, -1/2, infinity (rows: p^a; column 7: (0, 0, 0, 1)).
 *   in_mode: 0 fp32 x 16 pass-major (n, 4, H, W, 16) as written by out_mode 1; 1 collapsed first layer (x = cpos
 *   (H, W, 64), aux = (nimg, 25, 64), as tocvp_conv5x5_f32); 2 fp32 NHWC.
 *   out_mode: 0 fp32 NHWC; 1 fp32 x 16 pass-major; 2 fp16 operand planes of 2^8 y, pass-major (the planes input of
 *   tocvp_conv5x5_dec_f16x3_f32 / _tail_f32); 3 (tail_taps != NULL, else NULL) y = (nimg, 36, H, W) tap products of the
 *   folded decoder tail (tocvp_pack_tail_taps_f16x3, summed by tocvp_dec_tail_sum_f32).
 *   in_amax (in_mode 2, nullable): DEVICE word holding max |x| as tocvp_absmax_f32 writes it -- the input is then scaled by the
 *   largest power of two that keeps it inside the fp16 planes instead of by 16: inputs of any magnitude (the data gradients of
 *   the training step, 04_train_predictor.py); gate (out_mode 0, nullable): NHWC array, outputs are zeroed where gate <= 0
 *   (the ReLU of the layer below in the backward pass).
 * H % 4 == 0, W % 64 == 0.  Valid for |x| < 255 without in_amax (operands saturate beyond); no limit on the weights.
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_conv_weights_wino_f16x3_bytes(void);
int tocvp_split_conv_weights_wino_f16x3(const float* w, void* wf, const float* scales, float* absmax_out, int Cout,
                                        int Cin, void* stream);
int tocvp_conv5x5_dec_wino_f16x3_f32(const float* x, const float* aux, int in_mode, const void* wf, const float* coef,
                                     const float* bias, const void* tail_taps, float* y, int nimg, int H, int W,
                                     int relu, int out_mode, const float* in_amax, const float* gate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder tail folded into the last hidden layer.  The tail Conv2d(64 -> 4, k = 3) (decoders.py:112-117) is linear,
 * so the last 5 x 5 layer's epilogue multiplies every pixel's 64 outputs (bias + ReLU applied, still on the chip,
 * no halo) with the 36 x 64 tap matrix (rows 4 t + o, t = 3 ky + kx) in the same split-fp16 arithmetic and writes
 * `products` (nimg, 36, H, W) fp32 instead of the 64-channel activation; tocvp_dec_tail_sum_f32 then forms
 *     rgba[o](p) = bias[o] + sum_t products[4 t + o](p + off_t)            (zero outside the image)
 * the softmax of alpha over the K slot images of a frame and the composited frame -- the outputs of
 * tocvp_dec_tail_f32 (SAVi.py:251-255).  tail_taps: image written by tocvp_pack_tail_taps_f16x3 from the tail
 * weight (4, 64, 3, 3), tocvp_tail_taps_f16x3_bytes() bytes.  layout: bit 0 pass-major input, bit 3 operand planes in
 * it, as for tocvp_conv5x5_dec_f16x3_f32.  Valid for |activation| < 255, |tail weight| < 63 (saturating beyond).
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_tail_taps_f16x3_bytes(void);
int tocvp_pack_tail_taps_f16x3(const float* w, void* out, void* stream);
int tocvp_conv5x5_dec_f16x3_tail_f32(const float* x, const void* wf, const float* bias, const void* tail_taps,
                                     float* products, int nimg, int H, int W, int relu, int layout, void* stream);
int tocvp_dec_tail_sum_f32(const float* products, const float* bias, float* recons_imgs, float* recons, float* masks,
                           int F, int K, int H, int W, void* stream);
/* Placed form (round 4): frame f of the launch is written at recons_imgs + f * img_fs, recons + f * rec_fs,
 * masks + f * mask_fs (strides in floats, at least one frame each), so a per-step decode of B frames lands directly at
 * rows b * P + t of the evaluator's (B * P, ...) results (05_evaluate_predictor.py:88-96 reshapes them from one decode
 * call; here they are never stacked or copied).  clamped_imgs (nullable, same stride as recons_imgs) also receives
 * clamp(recons_imgs, 0, 1) -- the evaluator's pred_imgs (:93) -- with torch.clamp's NaN behaviour. */
int tocvp_dec_tail_sum_placed_f32(const float* products, const float* bias, float* recons_imgs, float* recons,
                                  float* masks, float* clamped_imgs, long img_fs, long rec_fs, long mask_fs, int F,
                                  int K, int H, int W, void* stream);

/* tap-sum matrices of the collapsed decoder layer 0:
 *   out[cls=(cy*5+cx), co, ci] = sum over taps (dy,dx) valid for border class (cy,cx) of
 *   w[co,ci,dy,dx];  w: (Cout,Cin,5,5), out: (25,Cout,Cin). */
int tocvp_dec_tapsum_f32(const float* w, float* out, int Cout, int Cin, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder tail (SAVi.py:251-255 + decoders.py:111-118): Conv2d(Cin->4,k3,p1) per slot image,
 * split RGB / alpha, softmax over the K slots, recon = sum_K rgb * mask.
 *   x:(F*K,H,W,Cin) NHWC; w:(4,Cin,3,3) nn.Conv2d layout; bias:(4)
 *   recons_imgs:(F,3,H,W)  recons:(F,K,3,H,W)  masks:(F,K,1,H,W)   (reference output layouts)
 *   K <= 32, Cin == 64, H % 16 == 0, W % 16 == 0.
 *   ws: >= 9*Cin*4*4 bytes of scratch (weights repacked to [tap][c][4] for scalar-path reads).
 * ------------------------------------------------------------------------------------------- */
int tocvp_dec_tail_f32(const float* x, const float* w, const float* bias, float* recons_imgs,
                       float* recons, float* masks, int F, int K, int H, int W, int Cin,
                       void* ws, size_t ws_bytes, void* stream);
/* placed form: output frame strides + optional clamped frames, as tocvp_dec_tail_sum_placed_f32 */
int tocvp_dec_tail_placed_f32(const float* x, const float* w, const float* bias, float* recons_imgs, float* recons,
                              float* masks, float* clamped_imgs, long img_fs, long rec_fs, long mask_fs, int F, int K,
                              int H, int W, int Cin, void* ws, size_t ws_bytes, void* stream);

/* dst (rows, row_len) contiguous = clamp(src, 0, 1) over rows of row_len floats that lie src_row_stride floats apart
 * (row_len, src_row_stride % 4 == 0, 16-byte aligned); NaN stays NaN like torch.clamp.  The evaluator's
 * targets = videos[:, ctx : ctx + P].clamp(0, 1) (05_evaluate_predictor.py:95) in one pass. */
int tocvp_clamp01_rows_f32(const float* src, long src_row_stride, float* dst, long rows, long row_len, void* stream);

/* dst[i0, i1, i2, 0:L] = src[i0, i1, i2, 0:L]: strided 4-d copy, strides in floats and independent on both sides; L, every
 * stride and both bases multiples of 4 floats.  The index copies of the hot path (torch.cat / torch.stack / .contiguous() of
 * window slices in predictor_wrapper.py:60-69 and text_cond_OCVP.py:96-113, the frame loop of SAVi.py:139-223) without a
 * torch kernel. */
int tocvp_copy4d_f32(const float* src, long ss0, long ss1, long ss2, float* dst, long ds0, long ds1, long ds2, int n0,
                     int n1, int n2, int L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text-encoder front end (text_encoders.py:89-103): token + position embedding, LayerNorm
 * (eps), zero the rows of padding tokens (id 0).  tokens: int64 (B,L); out: (B,L,D), D == 128.
 * ------------------------------------------------------------------------------------------- */
int tocvp_text_embed_f32(const int64_t* tokens, const float* tok_emb, const float* pos_emb,
                         const float* gamma, const float* beta, float* out, int B, int L, int D,
                         int vocab, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LearnedRandom slot initialiser (models/Blocks/initializers.py:87-94):
 *   out[r,:] = mu + sigma * noise[r,:],  mu/sigma: (D), noise/out: (rows, D).
 * The Gaussian draw itself stays on the host CPU generator for RNG parity with the reference.
 * ------------------------------------------------------------------------------------------- */
int tocvp_slot_init_f32(const float* mu, const float* sigma, const float* noise, float* out,
                        int rows, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * T5 text encoder pieces of TextOCVP_T5 (models/Predictors/text_cond_OCVP.py:141-151 ->
 * transformers.T5EncoderModel, un-pinned third-party dependency; its published algorithm restated):
 *  tocvp_rmsnorm_f32:   T5LayerNorm  y = x * rsqrt(mean(x^2) + eps) * gamma
 *  tocvp_embedding_f32: token embedding lookup out[r] = table[ids[r]]
 *  tocvp_mha_bias_f32:  tocvp_mha_f32 with an additive score bias (H, Tq, Tk) shared by the batch
 *                       (T5's bucketed relative-position bias; T5 uses scale = 1)
 * ------------------------------------------------------------------------------------------- */
int tocvp_rmsnorm_f32(const float* x, const float* gamma, float* y, int rows, int D, float eps,
                      void* stream);
int tocvp_embedding_f32(const int64_t* ids, const float* table, float* out, int rows, int D,
                        int vocab, void* stream);
int tocvp_mha_bias_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                       float* O, int ldo, int B, int H, int Tq, int Tk, int dh, float scale,
                       const int32_t* key_len, const float* bias, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ExtendedDINOSAUR decode side (models/EncodersDecoders/decoders.py:203-365):
 *  tocvp_conv3x3_f32: Conv2d(Cin->Cout,k3,p1) with per-channel scale/shift epilogue (BatchNorm in
 *    eval mode folded: scale = gamma/sqrt(var+eps), shift = (bias-mean)*scale+beta; scale may be
 *    NULL = 1) and optional ReLU, NHWC fp32, fp32 MFMA.  upsample2 != 0: the input is the nearest x2
 *    upsampling of x ((nimg,H/2,W/2,Cin)), fused into the tile loader (model_blocks.py:23-45).
 *    wp: (9, Cout, Cin) from tocvp_pack_conv_weights_f32.  Cin % 64 == 0, Cout % 32 == 0, H % 8 == 0.
 *  tocvp_slot_composite_f32: decoded (B,K,N,ld) with features in [0,F) and alpha at F (ld >= F+1: the
 *    producing GEMM may pad its width) -> recons (B,N,F) = sum_k feats*softmax_K(alpha), masks (B,K,N)
 *    (decoders.py:279-283).  K <= 64.
 *  tocvp_bilinear_resize_f32: F.interpolate(bilinear, align_corners=False) (decoders.py:291-297);
 *    x: NHWC with channel stride `cstride` (first C channels used), y: NCHW (n,C,OH,OW).
 * ------------------------------------------------------------------------------------------- */
int tocvp_conv3x3_f32(const float* x, const float* wp, const float* scale, const float* shift,
                      float* y, int nimg, int H, int W, int Cin, int Cout, int relu, int upsample2,
                      void* stream);
/* same convolution with f16x3 split operands (fp32-class, |x| < 255, |w| < 63; Cin % 32 == 0): 3/16 of the
 * matrix cycles of the fp32 MFMA form.  Round 5: ``wfrag`` = the fp16 operand planes of 2^10 w in MFMA-fragment order,
 * made ONCE per weight version by tocvp_split_weights_frag_f16(packed, wfrag, taps * Cout, Cin) on the packed
 * (taps, Cout, Cin) weights of tocvp_pack_conv_weights_f32 -- the kernel reads its weight fragments straight from L2 (rounds
 * 2-4 took the fp32 packed weights and split a tap's slice per workgroup behind a barrier per tap).  The same holds for
 * tocvp_conv3x3_up2_f16x3_f32 (packed: (4 phases x 4 taps, Cout, Cin)) and tocvp_conv5x5_f16x3_f32 ((25, Cout, Cin)). */
int tocvp_conv3x3_f16x3_f32(const float* x, const void* wfrag, const float* scale, const float* shift,
                      float* y, int nimg, int H, int W, int Cin, int Cout, int relu, int upsample2,
                      void* stream);
/* "Upsample(scale_factor=2) -> Conv2d(k3, p1)" of the image head (models/EncodersDecoders/decoders.py:325-365, the
 * Upsample of Blocks/model_blocks.py:23-45 in front of the next ConvBlock) as four 2x2 convolutions over the SOURCE image,
 * one per output phase (2 y + a, 2 x + b): the 3x3 taps that read the same source pixel are summed beforehand -- 16 tap
 * products per source pixel instead of 36 (2.25x fewer FLOPs), same result up to the fp32 rounding of the tap sums;
 * f16x3 arithmetic and scale / shift / ReLU epilogue of tocvp_conv3x3_f16x3_f32.
 *   x (nimg, SH, SW, Cin) NHWC; wphase (4 phases = 2 a + b, 4 taps = 2 i + j, Cout, Cin): tap (i, j) of phase (a, b)
 *   reads source pixel (y + i + a - 1, x + j + b - 1) and holds the sum of w[:, :, dy, dx] over dy in rows(a, i),
 *   dx in rows(b, j), rows(0, 0) = {0}, rows(0, 1) = {1, 2}, rows(1, 0) = {0, 1}, rows(1, 1) = {2};
 *   y (nimg, 2 SH, 2 SW, Cout).  Cin % 32 == 0, Cout % 32 == 0, SH % 8 == 0. */
int tocvp_conv3x3_up2_f16x3_f32(const float* x, const void* wphase_frag, const float* scale, const float* shift,
                                float* y, int nimg, int SH, int SW, int Cin, int Cout, int relu, void* stream);
/* generic 5x5 convolution (pad 2, + bias, optional ReLU; Cin % 32 == 0, Cout % 32 == 0, H % 8 == 0) with
 * the same f16x3 split operands: SAVi encoder convs 32 -> 32 (encoders.py:99-159).  wfrag: fragment-order planes of the
 * (25, Cout, Cin) weights of tocvp_pack_conv_weights_f32 (see tocvp_conv3x3_f16x3_f32). */
int tocvp_conv5x5_f16x3_f32(const float* x, const void* wfrag, const float* bias, float* y, int nimg,
                            int H, int W, int Cin, int Cout, int relu, void* stream);
int tocvp_slot_composite_f32(const float* decoded, float* recons, float* masks, int B, int K, int N,
                             int F, int ld, void* stream);
int tocvp_bilinear_resize_f32(const float* x, float* y, int n, int C, int cstride, int SH, int SW,
                              int OH, int OW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluation metrics that consume the rendered frames (reference lib/metrics.py:181-255, which
 * delegates to piqa==1.2.2 -- not vendored; the published definitions are restated):
 *   psnr[n] = 10 log10(1 / (mse_n + 1e-8)),   ssim[n] = mean SSIM map (11-tap Gaussian sigma 1.5,
 *   valid padding, k1 0.01, k2 0.03, channel average).  preds/targets: (N, C, H, W) fp32;
 *   clamp01 != 0 clamps both to [0,1] on load (the evaluator's clamp, 05_evaluate_predictor.py:96-99).
 *   psnr or ssim may be NULL.  ws: tocvp_metrics_ws_bytes(N, C) bytes.  H*W*8 <= 160 KiB.
 * ------------------------------------------------------------------------------------------- */
size_t tocvp_metrics_ws_bytes(int N, int C);
int tocvp_psnr_ssim_f32(const float* preds, const float* targets, float* psnr, float* ssim, int N,
                        int C, int H, int W, int clamp01, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text cross-attention of one predictor block, collapsed over the caption (csrc/xattn.hip):
 *     y = x + out_projection( softmax_t( (LayerNorm(x) Wq^T) K^T * scale ) V ) + bias
 * with the query / output projections folded into per-sample caption operands built once per rollout:
 *     G  (B * heads * 16, E): row (b, h, t) = sum_d Wq[h dh + d, :] K[b, t, h dh + d]   (zero rows for t >= Lt)
 *     HT (B * E, heads * 16): row (b, c), column (h, t) = sum_d Wo[c, h dh + d] V[b, t, h dh + d]
 * both as fp16 operand planes of 2^10 w in MFMA-fragment order (tocvp_split_weights_frag_f16 of the fp32 matrices).
 * x, y (B, Tq, E) fp32 contiguous; E = 512, heads = 8, dh = 64, 1 <= Lt <= 16.  One launch replaces LayerNorm,
 * nn.Linear q, MetaAttention.attention over the caption and out_projection (+ residual) of the reference's
 * TransformerDecoderBlock.forward (models/Blocks/attention.py:445-463, 303-319); f16x3 arithmetic (fp32-class),
 * LayerNorm and softmax in fp32.  Padded caption positions take part as in the reference (no key mask).
 * ------------------------------------------------------------------------------------------- */
int tocvp_xattn_collapsed_f32(const float* x, const float* gamma, const float* beta, float eps, const void* Gfrag,
                              const void* Hfrag, const float* bias, float* y, int B, int Tq, int E, int heads,
                              int Lt, float scale, void* stream);

/* =============================================================================================
 * Predictor TRAINING step (reference 04_train_predictor.py:57-108,
 * lib/loss.py:150-191, lib/setup_model.py:285-332): backward and optimiser kernels.  The forward
 * pass reuses the inference entry points above.
 * ============================================================================================= */

/* C[b1,b2] (M x N) = alpha * op(A[b1,b2]) (M x K) * op(B[b1,b2]) (K x N) (+ C if accumulate), exact fp32
 * MFMA; op = identity / transpose; ld* leading dimensions, s*1 / s*2 batch strides in elements.
 * Replaces autograd's nn.Linear backward (dW = dY^T X, dX = dY W) and the matmuls of attention backward
 * (MetaAttention.attention, attention.py:157-176, differentiated). */
int tocvp_bmm_f32(const float* A, int lda, long sA1, long sA2, int transA, const float* B, int ldb,
                  long sB1, long sB2, int transB, float* C, int ldc, long sC1, long sC2, int nb1,
                  int nb2, int M, int N, int K, float alpha, int accumulate, void* stream);
/* Weight gradient of nn.Linear without transposes: split z of `splits` reduces token rows
 * [z * chunk, (z + 1) * chunk) (chunk = ceil(M / splits) rounded up to 16) of
 *     c_part[z] (N x K) (+)= G[rows]^T X[rows],    bias_part[z] (N) (+)= column sums of G[rows]  (or NULL)
 * G (M x N, leading dimension ldg) = dL/dY, X (M x K, ldx) = the layer input; exact fp32 MFMA, operands read
 * as they lie (LDS-DMA).  M % 16 == 0, N % 128 == 0, K % 128 == 0, 16-byte aligned rows.  The caller sums
 * the splits (fixed order; tocvp_colsum_partial_f32) once per backward pass: every rollout step accumulates
 * into the same partial buffers.  Replaces autograd's dW = dY^T X / db = sum(dY) of nn.Linear
 * (04_train_predictor.py:96-104). */
int tocvp_gemm_tn_f32(const float* G, int ldg, const float* X, int ldx, float* c_part, float* bias_part,
                      int M, int N, int K, int splits, int accumulate, void* stream);
/* The same contract on the bf16 matrix cores with split operands: every value as two bf16 planes (16 significant
 * bits, fp32 exponent range), three products per algorithmic product (~2^-17 each) into fp32 accumulators; the tiles
 * stay row-major in LDS and both fragments are hardware-transposed reads (ds_read_b64_tr_b16).  M % 32 == 0 (the chunk
 * of a split is rounded up to 32 rows); bias sums are taken from the fp32 values (exact).  May be mixed with
 * tocvp_gemm_tn_f32 on the same partial buffers. */
int tocvp_gemm_tn_bf16x3_f32(const float* G, int ldg, const float* X, int ldx, float* c_part, float* bias_part,
                             int M, int N, int K, int splits, int accumulate, void* stream);
/* Round 5: the same product over SEVERAL row segments in ONE launch -- dW = sum_t G_t^T X_t of a weight that back-propagation
 * through time uses once per rollout step (04_train_predictor.py:57-108 unrolls the rollout un-detached).  G[s] (rows[s] x N),
 * X[s] (rows[s] x K), s < nseg <= 20, rows[s] % 32 == 0; G / X / rows are HOST arrays whose entries travel as kernel arguments
 * (nothing to keep alive or to re-upload under a captured HIP graph).  Partial buffers, splits and accumulate as above, over
 * the concatenated rows. */
int tocvp_gemm_tn_bf16x3_multi_f32(const float* const* G, const float* const* X, const int* rows, int nseg, int ldg, int ldx,
                                   float* c_part, float* bias_part, int N, int K, int splits, int accumulate, void* stream);
/* Backward of multi-head softmax attention, fused (scores / probabilities never reach HBM), exact fp32 MFMA:
 * q, o, d_o, dq (B, Tq, E); k, v, dk, dv (B, Tk, E); E = H * 64; o = the forward output; stats (B, H, Tq, 2) is
 * scratch (log-sum-exp and <dO, O> per query row); key_len (B) int32 or NULL masks keys >= key_len[b] (their
 * dk / dv rows are zero).  dq / dk / dv are overwritten.  Three launches (row statistics; dK, dV with a wave
 * per 32 keys; dQ with a wave per 32 queries), no atomics.  Replaces autograd through
 * MetaAttention.attention (attention.py:157-176) in the training step. */
int tocvp_attn_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                       float* dq, float* dk, float* dv, float* stats, const int32_t* key_len, int B, int H,
                       int Tq, int Tk, int E, float scale, void* stream);
/* y[r,:] = softmax(scale * x[r,:]) over the first key_len[r / rows_per_batch] (or all) columns, masked
 * columns 0 (recomputed probabilities of attention backward); ds = scale * p * (dp - <p, dp>). */
int tocvp_softmax_rows_f32(const float* x, float* y, int rows, int cols, float scale,
                           const int32_t* key_len, int rows_per_batch, void* stream);
int tocvp_softmax_bwd_f32(const float* p, const float* dp, float* ds, int rows, int cols, float scale,
                          void* stream);
/* y = act(x) and dx = dy * act'(x) for TOCVP_ACT_RELU / TOCVP_ACT_GELU (for ReLU x may be the output) */
int tocvp_act_f32(const float* x, float* y, long n, int act, void* stream);
int tocvp_act_bwd_f32(const float* dy, const float* x, float* dx, long n, int act, void* stream);
/* nn.Dropout (text encoder, text_encoders.py:36,64,107 and its nn.TransformerEncoderLayer): the caller
 * supplies the uniform sample r in [0,1); y = r >= p ? x / (1 - p) : 0; backward = same call on the gradient */
int tocvp_dropout_f32(const float* x, const float* r, float* y, long n, float p, void* stream);
/* y = a * x + b * y (gradient accumulation, residual adds) */
int tocvp_axpby_f32(const float* x, float* y, long n, float a, float b, void* stream);
/* partial[chunk, c] = sum of x[r, c] over the rows of the chunk (bias / parameter gradients; run once
 * more on `partial` with one chunk for the total) */
int tocvp_colsum_partial_f32(const float* x, float* partial, int rows, int cols, int ld,
                             int rows_per_chunk, void* stream);
/* nn.LayerNorm backward: dx (rows, D) and per-wave partial sums pgamma / pbeta (nwaves, D), nwaves % 4 == 0,
 * written, or added to what the rows already hold (accumulate bit 0: one partial buffer per parameter
 * collects every use inside a backward pass and is column-summed once); accumulate bit 1: dx += instead of
 * dx = (the input already carries the gradient of a residual branch) */
int tocvp_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, float* dx,
                            float* pgamma, float* pbeta, int nwaves, int rows, int D, float eps,
                            int accumulate, void* stream);
/* dW[ids[i], :] += dy[i, :]  (nn.Embedding backward; ids < 0 skipped) */
int tocvp_embedding_bwd_f32(const int64_t* ids, const float* dy, float* dW, int n, int D, void* stream);
/* nn.MSELoss: partial[b] = block sums of (pred - target)^2 (nblocks of them), dpred = gscale * (pred - target)
 * when dpred != NULL */
int tocvp_mse_f32(const float* pred, const float* target, float* partial, int nblocks, float* dpred,
                  long n, float gscale, void* stream);
/* partial[b] = block sums of x^2 (clip_grad_norm_) */
int tocvp_sqnorm_partial_f32(const float* x, float* partial, int nblocks, long n, void* stream);
/* Frozen SAVi decoder, backward w.r.t. the slots (image-loss term; SAVi.py:241-275 differentiated):
 *  tocvp_dec_tail_grad_f32: dimg (F,3,H,W), recons (F,K,3,H,W), masks (F,K,1,H,W) -> dy (F*K,H,W,4) =
 *    gradient of the tail conv's output (rgb | alpha) through compositing and the softmax over slots;
 *  tocvp_conv3x3_t4_f32: dx (n,H,W,C) = relu'(act) * conv_transpose3x3(dy (n,H,W,4), w (4,C,3,3));
 *  tocvp_dec_class_reduce_f32: dS (n,25,64) = per-border-class sums of g (n,H,W,64) * (cpos + S > 0)
 *    (collapsed decoder layer 0, see tocvp_dec_tapsum_f32).
 * The 5x5 data gradients in between reuse tocvp_conv5x5_bf16x3_f32 with transposed, flipped weights. */
int tocvp_dec_tail_grad_f32(const float* dimg, const float* recons, const float* masks, float* dy, int F,
                            int K, int H, int W, void* stream);
int tocvp_conv3x3_t4_f32(const float* dy, const float* w, const float* act, float* dx, int nimg, int H,
                         int W, int C, void* stream);
int tocvp_dec_class_reduce_f32(const float* g, const float* cpos, const float* S, float* dS, int nimg,
                               int H, int W, int C, void* stream);
/* torch.optim.Adam step (no weight decay / amsgrad) on a flat parameter.  The step-dependent scalars are
 * read from DEVICE memory so that a captured HIP graph of the training step can be replayed:
 * hyper = {lr, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t}; gscale (may be NULL) = clipping factor.
 * tocvp_clip_scale_f32: out[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) (clip_grad_norm_), out[1] = norm */
int tocvp_adam_f32(float* p, const float* g, float* m, float* v, long n, const float* hyper,
                   const float* gscale, void* stream);
int tocvp_clip_scale_f32(const float* sumsq, float max_norm, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TOCVP_H */
