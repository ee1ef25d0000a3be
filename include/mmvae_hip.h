/*
 * mmvae_hip.h — C ABI of the MI355X (gfx950) hot-path library for the multimodal-VAE training step.
 *
 * The reference (gabinsane/multimodal-vae-comparison) has no native code and no FFI: every "kernel" it
 * runs is a stock PyTorch op reached through torch.nn.  The entry points below are the op
 * sequences of its per-step path, each citing the reference lines it replaces.  Paths are
 * relative to /root/reference/multimodal_compare/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated; no torch types, no allocation,
 *     no host synchronisation; every call only enqueues work on `stream` (graph-capturable, re-entrant
 *     per stream);
 *   - return value: 0 = ok, MMVAE_ERR_* otherwise (the Python binding raises RuntimeError);
 *   - `ws` = caller-provided workspace; the number of floats needed is returned by the matching
 *     *_ws_floats() query (0 => may be NULL);
 *   - `accumulate` != 0: results are ADDED to the destination (gradient accumulation into a flat grad
 *     buffer), else they overwrite it.
 */
#ifndef MMVAE_HIP_H
#define MMVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mmvae_stream_t; /* hipStream_t */

enum {
  MMVAE_OK = 0,
  MMVAE_ERR_ARG = 1,         /* null pointer / non-positive size */
  MMVAE_ERR_UNSUPPORTED = 2, /* shape outside what the kernels are built for */
  MMVAE_ERR_LAUNCH = 3       /* hipGetLastError() != hipSuccess after launch */
};

/* input transform applied to an operand while it is staged into LDS */
enum { MMVAE_ACT_NONE = 0, MMVAE_ACT_SILU = 1, MMVAE_ACT_RELU = 2, MMVAE_ACT_GELU = 3 };
/* epilogue applied to the accumulator before the store; `aux` has the output's shape */
enum {
  MMVAE_EP_NONE = 0,
  MMVAE_EP_RELU = 1,          /* y = max(acc + bias, 0)                                           */
  MMVAE_EP_MUL_RELU_MASK = 2, /* y = acc * (aux > 0)        (dgrad through a ReLU, aux = saved u or y) */
  MMVAE_EP_MUL_SILU_GRAD = 3, /* y = acc * silu'(aux)       (dgrad through a SiLU, aux = saved u)   */
  MMVAE_EP_GELU = 4,          /* y = gelu(acc + bias); aux (if non-null) RECEIVES acc + bias        */
  MMVAE_EP_MUL_GELU_GRAD = 5, /* y = acc * gelu'(aux)                                              */
  MMVAE_EP_SIGMOID_CLAMP = 6, /* y = clamp(sigmoid(acc + bias), 1e-6, 1 - 1e-6)  (Dec_CNN, decoders.py:96-97) */
  MMVAE_EP_SIGMOID = 7,       /* y = sigmoid(acc + bias)                          (Dec_SVHN, decoders.py:144) */
  MMVAE_EP_ADD_AUX = 8        /* y = acc + bias + aux   (dgrad of a sub-layer's first op + the residual branch's gradient) */
};

/* Inverted dropout (train mode of the text towers: nn.Dropout(0.1) in PositionalEncoding and in every
 * Transformer sub-layer, models/encoders.py:790, decoders.py:669).  Counter-based: element i of site `site` is kept
 * iff hash(seed, counter, site, i) >= p, so the backward kernels regenerate the mask instead of storing it.
 * state[0] = seed, state[1] = running counter, state[2 + slot] = the counter value of forward call `slot` of this
 * step (mmvae_dropout_advance bumps the counter and fills the slot; forward and backward of that call read it).
 * A NULL pointer / p == 0 disables dropout.  Philox streams of the reference cannot be matched bit for bit; parity
 * under dropout is tested with the masks extracted by mmvae_dropout_mask and fed to the oracle. */
typedef struct {
  const uint32_t* state;
  uint32_t slot;
  uint32_t site;
  float p;
} mmvae_dropout_t;
#define MMVAE_DROPOUT_SLOTS 16
int mmvae_dropout_advance(uint32_t* state, uint32_t slot, mmvae_stream_t stream);
/* mmvae_dropout_advance(states[i], 0) for n distinct towers in ONE launch: forward call 0 of every tower of a training
 * step, issued before the towers fork onto their streams */
#define MMVAE_DROPOUT_ADVANCE_MAX 16
int mmvae_dropout_advance_many(uint32_t* const* states, int n, mmvae_stream_t stream);
/* out[i] = 0 or 1/(1-p): the multiplicative mask of element i (test / inspection helper) */
int mmvae_dropout_mask(const mmvae_dropout_t* drop, float* out, long n, mmvae_stream_t stream);
/* y = dropout(act(x)) elementwise; dx = dy * mask * act'(x).  act: MMVAE_ACT_NONE or MMVAE_ACT_GELU */
int mmvae_dropout_act_fwd(const float* x, float* y, long n, int act, const mmvae_dropout_t* drop,
                          mmvae_stream_t stream);
int mmvae_dropout_act_bwd(const float* dy, const float* x, float* dx, long n, int act, const mmvae_dropout_t* drop,
                          mmvae_stream_t stream);
/* cross-attention over a length-1 memory with attention-weight dropout (nn.MultiheadAttention dropout on the
 * (N*H, L, 1) weights): out[l,n,c] = v[n,c] * mask[(n*H + head(c))*L + l];  bwd: dv[n,c] = sum_l dout * mask */
int mmvae_head_bcast_dropout_fwd(const float* v, float* out, int L, int N, int H, int hd, const mmvae_dropout_t* drop,
                                 mmvae_stream_t stream);
int mmvae_head_bcast_dropout_bwd(const float* dout, float* dv, int L, int N, int H, int hd,
                                 const mmvae_dropout_t* drop, mmvae_stream_t stream);

/* Feed-forward block of nn.TransformerEncoderLayer / DecoderLayer with d_model = 32 (the action towers, reference
 * models/encoders.py:706-716, models/decoders.py:589-600) without materialising the (M, FF) hidden activation:
 *   y = W2 dropout(gelu(W1 x + b1)) + b2;   x, y (M,32); w1 (FF,32), b1 (FF), w2 (32,FF), b2 (32); FF % 32 == 0.
 * The dropout mask of hidden element (row, col) is that of mmvae_dropout_act_fwd at index row*FF + col (drop NULL or
 * p == 0: none).  Backward recomputes the hidden tiles: dx (M,32; may be NULL) and mmvae_ffn32_bwd_parts(M, FF) partial
 * rows of mmvae_ffn32_bwd_rowlen(FF) floats in ws, each [dW1 (FF,32) | db1 (FF) | dW2 (32,FF) | db2 (32)], to be summed
 * by the caller (mmvae_reduce_rows / mmvae_reduce_segments).  ws may be NULL too (data gradient only): the data- and
 * the weight-gradient launch are independent and may go to different streams. */
int mmvae_ffn32_supported(int d_model, int FF);
/* out_proj + dropout + residual + LayerNorm of a d_model-32 layer in one launch (reference models/encoders.py:706-716:
 * `src = norm1(src + dropout1(attn))`, the attention's out_proj being nn.Linear(32, 32)):
 *   y = LayerNorm(dropout(x W^T + b) + r);  x, r, y, xhat (M, 32), w (32, 32) [out][in], rstd (M); all 16-byte aligned.
 * Same saved tensors and dropout mask (element row * 32 + column) as mmvae_layernorm_residual_fwd on the Linear's output:
 * the backward is mmvae_layernorm_residual_bwd followed by the Linear's backward. */
int mmvae_proj32_ln_fwd(const float* x, const float* w, const float* b, const float* r, const float* gamma, const float* beta,
                        float* y, float* xhat, float* rstd, int M, const mmvae_dropout_t* drop, mmvae_stream_t stream);
int mmvae_ffn32_bwd_parts(int M, int FF);
size_t mmvae_ffn32_bwd_rowlen(int FF);
int mmvae_ffn32_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, int M,
                    int FF, const mmvae_dropout_t* drop, mmvae_stream_t stream);
int mmvae_ffn32_bwd(const float* x, const float* dy, const float* w1, const float* b1, const float* w2, float* dx,
                    float* ws, int M, int FF, const mmvae_dropout_t* drop, mmvae_stream_t stream);
/* The same three launches on split-bf16 MFMA (every fp32 operand = the exact sum of three bf16 terms, six bf16 MFMAs per
 * product, fp32 accumulation: the contract of the fp32 kernels above to ~3e-7 of the tensor maximum).  The weights are
 * split ONCE per step into wsplit (mmvae_ffn32_wsplit_bytes(FF) bytes, 16-byte aligned; mmvae_ffn32_prep_weights) and
 * that image replaces w1 / w2 in the calls; the weight-gradient launch also needs rsplit (mmvae_ffn32_rsplit_bytes(M)
 * bytes of scratch, written and read by that call in stream order).  Same partial-row layout in ws, same dropout mask.
 * dx_add (M,32) or NULL is added to dx (the gradient of the residual connection around the block). */
size_t mmvae_ffn32_wsplit_bytes(int FF);
size_t mmvae_ffn32_rsplit_bytes(int M);
int mmvae_ffn32_prep_weights(const float* w1, const float* w2, void* wsplit, int FF, mmvae_stream_t stream);
/* ... of up to 16 layers (host arrays of n device pointers each) in ONE launch */
int mmvae_ffn32_prep_weights_many(const float* const* w1, const float* const* w2, void* const* wsplit, int n, int FF,
                                  mmvae_stream_t stream);
int mmvae_ffn32_fwd_b16(const float* x, const void* wsplit, const float* b1, const float* b2, float* y, int M, int FF,
                        const mmvae_dropout_t* drop, mmvae_stream_t stream);
/* ... with the LayerNorm that consumes the block as the launch's epilogue: y = LayerNorm(dropout(ffn(x)) + r), xhat (M, 32)
 * and rstd (M) saved as mmvae_layernorm_residual_fwd saves them (ln_drop: the dropout in front of the residual sum, mask of
 * element row * 32 + column); backward = mmvae_layernorm_residual_bwd, then mmvae_ffn32_bwd_b16.
 * (reference models/encoders.py:706-716: `src = norm2(src + dropout2(linear2(dropout(activation(linear1(src))))))`) */
int mmvae_ffn32_fwd_b16_ln(const float* x, const void* wsplit, const float* b1, const float* b2, const float* r,
                           const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int M, int FF,
                           const mmvae_dropout_t* drop, const mmvae_dropout_t* ln_drop, mmvae_stream_t stream);
int mmvae_ffn32_bwd_b16(const float* x, const float* dy, const void* wsplit, const float* b1, float* dx, float* ws,
                        void* rsplit, const float* dx_add, int M, int FF, const mmvae_dropout_t* drop,
                        mmvae_stream_t stream);

int mmvae_version(void);
const char* mmvae_arch(void); /* "gfx950" */

/* ------------------------------------------------------------------------------------------------
 * 4x4 / stride 2 / padding 1 convolutions, NCHW fp32, square maps  (a2, a3 of SURVEY 8(a))
 *   conv2d      : torch.nn.Conv2d(k=4,s=2,p=1)          models/encoders.py:186-191,214-217
 *   convT2d     : torch.nn.ConvTranspose2d(k=4,s=2,p=1)  models/decoders.py:62-69,91-95
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 / 16x16x4_f32, im2col tiles staged through LDS.
 * Supported channel pairs: (3|32) -> 32 for the "gather" form, 32 -> (32|3) for the "scatter" form.
 * ---------------------------------------------------------------------------------------------- */

/* y[b,o,oh,ow] = ep( bias[o] + sum_{c,kh,kw} act(x[b,c,2oh-1+kh,2ow-1+kw]) * w[o,c,kh,kw] )
 * x (B,Cin,Hin,Hin) -> y (B,Cout,Hin/2,Hin/2); w (Cout,Cin,4,4).  bias/aux may be NULL. */
int mmvae_conv2d_k4s2_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y,
                          int B, int Cin, int Cout, int Hin, int in_act, int ep_mode, mmvae_stream_t stream);

/* dx[b,c,ih,iw] = ep( sum_{o,kh,kw: ih=2oh-1+kh, iw=2ow-1+kw} dy[b,o,oh,ow] * w[o,c,kh,kw] )
 * dy (B,Cout,Hout,Hout) -> dx (B,Cin,2Hout,2Hout); w (Cout,Cin,4,4).  ep: MUL_* with aux = saved input. */
int mmvae_conv2d_k4s2_dgrad(const float* dy, const float* w, const float* aux, float* dx,
                            int B, int Cin, int Cout, int Hout, int ep_mode, mmvae_stream_t stream);

/* dw[o,c,kh,kw] (+)= sum_{b,oh,ow} dy[b,o,oh,ow] * act(x[b,c,2oh-1+kh,2ow-1+kw]);  db[o] (+)= sum dy.
 * ws: mmvae_conv_wgrad_ws_floats(B,Cin,Cout,Hout) floats. */
int mmvae_conv2d_k4s2_wgrad(const float* dy, const float* x, float* dw, float* db, float* ws,
                            int B, int Cin, int Cout, int Hout, int x_act, int accumulate, mmvae_stream_t stream);

/* y[b,o,oh,ow] = ep( bias[o] + sum_{c,kh,kw: oh=2ih-1+kh, ow=2iw-1+kw} act(x[b,c,ih,iw]) * w[c,o,kh,kw] )
 * x (B,Cin,Hin,Hin) -> y (B,Cout,2Hin,2Hin); w (Cin,Cout,4,4). */
int mmvae_convT2d_k4s2_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y,
                           int B, int Cin, int Cout, int Hin, int in_act, int ep_mode, mmvae_stream_t stream);

/* dx[b,c,ih,iw] = ep( sum_{o,kh,kw} dy[b,o,2ih-1+kh,2iw-1+kw] * w[c,o,kh,kw] );  dy (B,Cout,2Hin,2Hin) */
int mmvae_convT2d_k4s2_dgrad(const float* dy, const float* w, const float* aux, float* dx,
                             int B, int Cin, int Cout, int Hin, int ep_mode, mmvae_stream_t stream);

/* dw[c,o,kh,kw] (+)= sum_{b,ih,iw} act(x[b,c,ih,iw]) * dy[b,o,2ih-1+kh,2iw-1+kw];  db[o] (+)= sum dy. */
int mmvae_convT2d_k4s2_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws,
                             int B, int Cin, int Cout, int Hin, int x_act, int accumulate, mmvae_stream_t stream);

size_t mmvae_conv_wgrad_ws_floats(int B, int Csmall, int Clarge, int Hsmall);

/* Whole-layer backward in ONE launch (input-gradient and weight-gradient workgroups side by side):
 *   conv2d : dx = dgrad(dy, w) * act'(x);  dw (+)= wgrad(dy, act(x));  db (+)= sum dy      x (B,Cin,2Hout,2Hout)
 *   convT2d: dx = dgrad(dy, w) * act'(x);  dw (+)= wgrad(act(x), dy);  db (+)= sum dy      dy (B,Cout,2Hin,2Hin)
 * ws as for the *_wgrad entry points; shapes that are not fused fall back to two launches. */
int mmvae_conv2d_k4s2_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                          int B, int Cin, int Cout, int Hout, int x_act, int accumulate, mmvae_stream_t stream);
int mmvae_convT2d_k4s2_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                           int B, int Cin, int Cout, int Hin, int x_act, int accumulate, mmvae_stream_t stream);
/* Dec_CNN's last layer and its reconstruction loss in ONE launch (round 5; csrc/conv_t3.inc):
 *   logits = convT2d(act(x), w, bias)  x (B,32,32,32) -> (B,3,64,64)       models/decoders.py:69,95
 *   x_hat  = clamp(sigmoid(logits), 1e-6, 1 - 1e-6)                         models/decoders.py:96-97   (never stored)
 *   row[b] = sum bce(x_hat[b], target[b])   (logs clamped at -100)          models/objectives.py:392-406
 *   dlogit = seed * (x_hat - target) where the clamp is inactive, else 0    (the term's ELBO weight is known: ops.ConstSeed)
 * part: (B, mmvae_convT3_bce_strips(B)) floats and ticket: (B) unsigned, zeroed once by the caller (the kernel leaves the
 * tickets zero), when an image is split over several workgroups (strips > 1); both may be NULL when strips == 1. */
int mmvae_convT3_bce_seeded(const float* x, const float* w, const float* bias, const float* target, float* row,
                            float* dlogit, float* part, unsigned* ticket, int B, int in_act, float seed,
                            mmvae_stream_t stream);
int mmvae_convT3_bce_strips(int B);
/* The 32-channel layers on 16x16 / 32x32 maps and the 3-channel image layers run on one of two GEMM cores from a
 * tile-count threshold on: "split-bf16" (every fp32 operand split EXACTLY into three bf16 terms, six bf16 MFMAs per
 * product, fp32 accumulate; dropped terms <= 3 * 2^-24 |a b|, i.e. fp32-equivalent: tests/test_hip_ops.py holds it to
 * 2e-6 of the tensor maximum against fp64) or fp32 MFMA.  split_bf16 = 1 (default) / 0 selects, < 0 keeps; returns the
 * previous setting.  Non-finite inputs: see DESIGN.md "split-bf16 and non-finite values". */
int mmvae_conv_plan(int split_bf16);

/* ------------------------------------------------------------------------------------------------
 * Small dense layers: torch.nn.Linear and the projections inside nn.MultiheadAttention /
 * nn.Transformer*Layer  (models/encoders.py:194,43-54,825-826; models/decoders.py:58-60,86-88,705)
 * One strided fp32 MFMA GEMM:  C[M,N] = ep( bias + act(A)[M,K] * B[K,N] )
 *   A(m,k) at A[m*sam + k*sak],  B(k,n) at Bm[k*sbk + n*sbn],  C row-major with leading dim ldc.
 *   bias: per column n, or NULL.  a_rowsum (optional, length M): (+)= sum_k act(A)(m,k)
 *   (the bias gradient when A = dy^T).  splitk > 1 needs ws of mmvae_gemm_ws_floats() floats and
 *   allows neither bias nor epilogue (partials are summed by mmvae_reduce_rows).
 * ---------------------------------------------------------------------------------------------- */
int mmvae_gemm_f32(const float* A, const float* Bm, const float* bias, const float* aux, float* C,
                   float* a_rowsum, float* ws, int M, int N, int K, long sam, long sak, long sbk, long sbn,
                   long ldc, int a_act, int b_act, int ep_mode, int accumulate, int splitk,
                   mmvae_stream_t stream);
size_t mmvae_gemm_ws_floats(int M, int N, int splitk);
/* number of partial results (rows of M*N [+ M] floats in ws) mmvae_gemm_f32 writes for a requested splitk */
int mmvae_gemm_splits(int M, int N, int K, int splitk);
/* Runtime switch of the LDS-tiled split-bf16 kernels that take wide layers (N, K >= 128) at 2048 .. 4096 rows over from the
 * fp32-MFMA bodies in mmvae_gemm_f32 / mmvae_linear_fwd / _bwd_data / _bwd (csrc/gemm_b16.inc; same fp32 contract, every
 * product from six bf16 MFMAs on three exact bf16 terms per operand).  Returns the previous setting; the environment
 * variable MMVAE_GEMM_B16=0 starts with it off.  (nn.Linear: reference models/encoders.py:196-200, models/decoders.py:58-62.) */
int mmvae_gemm_b16_set(int on);
/* grouped bias of a layer whose N outputs are C channels x G positions (nn.ConvTranspose2d on a 1x1 input run as a
 * GEMM, models/decoders.py:116,129-131):  y[r, c*G + g] += bias[c];   db[c] (+)= sum_r sum_g dy[r, c*G + g] */
int mmvae_bias_group_add(float* y, const float* bias, int rows, int C, int G, mmvae_stream_t stream);
size_t mmvae_bias_group_ws_floats(int rows, int C);   /* = parts * C; parts = mmvae_bias_group_parts(rows) */
int mmvae_bias_group_parts(int rows);
/* ws: mmvae_bias_group_ws_floats floats of row-block partials (parts, C); accumulate = MMVAE_ACC_DEFER leaves them
 * there for the caller's fold */
int mmvae_bias_group_grad(const float* dy, float* db, float* ws, int rows, int C, int G, int accumulate,
                          mmvae_stream_t stream);

/* y = ep(x W^T + b): x (M,K) ld ldx, W (N,K), y (M,N).  F.linear */
int mmvae_linear_fwd(const float* x, const float* w, const float* b, float* aux, float* y, int M, int N, int K,
                     long ldx, int x_act, int ep_mode, mmvae_stream_t stream);
/* dx = ep(dy W): dy (M,N), W (N,K), dx (M,K) */
int mmvae_linear_bwd_data(const float* dy, const float* w, const float* aux, float* dx, int M, int N, int K,
                          int ep_mode, int accumulate, mmvae_stream_t stream);
/* dW (+)= dy^T act(x), db (+)= colsum(dy); dW (N,K).  ws: mmvae_linear_bwd_weight_ws_floats(M,N,K) */
int mmvae_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, float* ws, int M, int N, int K,
                            long ldx, int x_act, int accumulate, mmvae_stream_t stream);
size_t mmvae_linear_bwd_weight_ws_floats(int M, int N, int K);
/* Several independent weight gradients (the (L*N)-row ones a fused text layer leaves behind) in ONE launch: every
 * job gets exactly the tiling / split plan / partial layout of mmvae_linear_bwd_weight, so results are bit-identical to
 * n_jobs separate calls -- which is also what happens when a job falls outside the shared-grid regime (the
 * register-operand body; split jobs need accumulate == MMVAE_ACC_DEFER) or n_jobs > MMVAE_WGRAD_BATCH_MAX.
 * Replaces the autograd-generated per-Linear weight-gradient matmuls of nn.TransformerEncoderLayer /
 * nn.TransformerDecoderLayer.backward (reference models/encoders.py:818-824, models/decoders.py:698-704). */
#define MMVAE_WGRAD_BATCH_MAX 8
typedef struct {
  const float* dy; const float* x; float* dw; float* db; float* ws;
  int M, N, K; long ldx; int x_act, accumulate;
} mmvae_wgrad_job_t;
int mmvae_linear_bwd_weight_batch(const mmvae_wgrad_job_t* jobs, int n_jobs, mmvae_stream_t stream);
/* All weight gradients behind one fused text layer in ONE launch (csrc/twgrad.hip; round 4): dW_j = dy_j^T x_j and
 * db_j = colsum(dy_j) of torch.nn.TransformerEncoderLayer / TransformerDecoderLayer's linears
 * (models/encoders.py:828-837, models/decoders.py:708-723), each operand row fetched once.  Every job writes
 * nz = mmvae_txt_wgrad_splits(M, N, K) partial rows to ws, row z at ws + z * pitch = [N * K weight sums | N bias sums]
 * with pitch = mmvae_txt_wgrad_ws_floats(M, N, K) / nz (N * K + N rounded up to even), to be folded by
 * mmvae_reduce_segments / mmvae_adam_fold_flat -- as ONE segment of length N * K + N when weight and bias are adjacent in
 * the gradient buffer.  N (K) must be even when > 32. */
typedef struct {
  const float* dy; const float* x; float* ws;
  int M, N, K;
} mmvae_txt_wgrad_job_t;
int mmvae_txt_wgrad(const mmvae_txt_wgrad_job_t* jobs, int n_jobs, mmvae_stream_t stream);
int mmvae_txt_wgrad_splits(int M, int N, int K);
size_t mmvae_txt_wgrad_ws_floats(int M, int N, int K);
int mmvae_txt_wgrad_supported(int M, int N, int K);
/* both of the above in one grouped launch: dx = ep(dy W), dW (+)= dy^T act(x), db (+)= colsum(dy) */
int mmvae_linear_bwd(const float* dy, const float* x, const float* w, const float* aux, float* dx, float* dw,
                     float* db, float* ws, int M, int N, int K, long ldx, int x_act, int ep_mode, int accumulate,
                     mmvae_stream_t stream);
size_t mmvae_linear_bwd_ws_floats(int M, int N, int K);

/* ------------------------------------------------------------------------------------------------
 * Encoder heads: VaeComponent.process_output, models/encoders.py:49-54
 *   h (B, 2D) = [mu | pre-softmax]  ->  lv = softmax(h[:, D:], -1) + 1e-6, written in place.
 * bwd: dh[:, D:] = softmax backward of dlv (in place on dh, which holds [dmu | dlv]).
 * ---------------------------------------------------------------------------------------------- */
int mmvae_head_softmax_fwd(float* h, int B, int D, mmvae_stream_t stream);
int mmvae_head_softmax_bwd(const float* h, float* dh, int B, int D, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused latent op: product of experts -> reparameterised samples -> analytic KL
 *   TorchMMVAE.product_of_experts   models/mmvae_base.py:203-222   (returns VARIANCE, used as sigma)
 *   MoPOE.poe_fusion / forward      models/mmvae_models.py:385-394,351-370
 *   POE.modality_mixing / forward   models/mmvae_models.py:189-232
 *   weighted_group_kld / kl_normal  models/objectives.py:184-201, utils.py:399-402
 *   model prior sigma = softmax(theta)*D  models/mmvae_models.py:274-276
 *
 * experts: E pointers (device array of E device pointers is NOT used; pass up to 8 pointers by value)
 *   mu[e], lv[e] : (B,D) each with row stride ld_in (2D when they are the halves of a packed head output;
 *                  dmu/dlv use the same stride); lv is the encoder's softmax output, used as log-variance
 *                  inside PoE.  eps, z, dz are contiguous (B,D).
 *   with_prior   : 0 = product of the experts only; 1 = add the N(0,1) expert (mu 0, logvar 0);
 *                  2 = no product (E must be 1): the "joint" is expert 0 itself with sigma = its lv -- the
 *                  per-modality posteriors of MoE (models/mmvae_models.py:96-100).
 *   n_z draws    : z[i] = mu_J + var_J * eps[i]   (eps[i], z[i] : (B,D))
 *   kl (n_kl,B)  : row j < E: sum_d KL(N(mu_j, sigma=lv_j) || p) if kl_mask bit j set;
 *                  row E    : sum_d KL(N(mu_J, sigma=var_J) || p) if kl_mask bit E set; p = N(0, softmax(theta)*D)
 *   joint (2,B,D): mu_J, var_J (always written).
 * bwd: dz[i] (B,D), dkl (E+1,B)  ->  dmu[e], dlv[e] (B,D), dtheta (D) (+)=; ws: mmvae_poe_ws_floats(B,D).
 *   raw_heads    : != 0: lv[e] points at the RAW logvar-head outputs u and the kernels apply
 *                  lv = softmax(u, -1) + 1e-6 themselves (process_output, encoders.py:52-53), forward and backward
 *                  (dlv[e] then receives d/du) -- no mmvae_head_softmax_* launch per tower and direction.
 * ---------------------------------------------------------------------------------------------- */
#define MMVAE_MAX_EXPERTS 8
typedef struct {
  const float* mu[MMVAE_MAX_EXPERTS];
  const float* lv[MMVAE_MAX_EXPERTS];
  const float* eps[MMVAE_MAX_EXPERTS];
  float* z[MMVAE_MAX_EXPERTS];
} mmvae_poe_fwd_args;
typedef struct {
  const float* mu[MMVAE_MAX_EXPERTS];
  const float* lv[MMVAE_MAX_EXPERTS];
  const float* eps[MMVAE_MAX_EXPERTS];
  const float* dz[MMVAE_MAX_EXPERTS];
  float* dmu[MMVAE_MAX_EXPERTS];
  float* dlv[MMVAE_MAX_EXPERTS];
} mmvae_poe_bwd_args;
/* rng_state: NULL (a->eps are inputs), or the mmvae_randn generator state: the n_z draws are then generated by this
 * launch -- element i*B*D + b*D + d of the current draw, the values mmvae_randn would put into a (n_z,B,D) tensor --
 * written to a->eps[i] (outputs, kept for the backward pass), and the generator advances by one draw */
int mmvae_poe_reparam_kl_fwd(const mmvae_poe_fwd_args* a, const float* theta, float* joint, float* kl, int E,
                             int with_prior, int n_z, unsigned kl_mask, int B, int D, int ld_in, int raw_heads,
                             uint32_t* rng_state, mmvae_stream_t stream);
/* ticket: NULL, or a zero-initialised device int owned by the calling stream: the prior-parameter gradient is then
 * folded by the last workgroup of the same launch (the kernel leaves the int at zero) instead of a second launch */
int mmvae_poe_reparam_kl_bwd(const mmvae_poe_bwd_args* a, const float* theta, const float* dkl, float* dtheta,
                             float* ws, int* ticket, int E, int with_prior, int n_z, unsigned kl_mask, int B, int D,
                             int ld_in, int raw_heads, int accumulate, mmvae_stream_t stream);
/* ... acc_packed bit e: dmu[e] / dlv[e] already hold another call's contribution to these columns and this one is ADDED (the
 * same head output read by several fusion calls -- DMVAE's joint, shared and private posteriors, models/mmvae_models.py:
 * 480-502 -- accumulates its gradient in one tensor instead of one tensor per call plus autograd's addition launches).
 * Not with raw_heads. */
int mmvae_poe_reparam_kl_bwd_acc(const mmvae_poe_bwd_args* a, const float* theta, const float* dkl, float* dtheta,
                                 float* ws, int* ticket, int E, int with_prior, int n_z, unsigned kl_mask, int B, int D,
                                 int ld_in, int raw_heads, int accumulate, unsigned acc_packed, mmvae_stream_t stream);
size_t mmvae_poe_ws_floats(int B, int D);

/* Latent samples -> decoder inputs, and the transpose (round 6).  The reference decodes every subset's sample in a call of its
 * own (POE.objective, models/mmvae_models.py:159-187) and builds DMVAE's decoder inputs with torch.cat([z_shared, z_private], -1)
 * per pass (:494-502); here a decoder's passes are one batch, and the batches of ALL decoders are written by one launch:
 * block k copies B rows of `width[k]` floats, dst[k][b * ld_dst[k] + d] = src[k][b * ld_src[k] + d] (dst already points at the
 * block's first row / column of its output).  bwd: out[s] (B, width[s]) = sum_j g[s][j][b * ld[s][j] + d] over the n_g[s] <= 4
 * blocks that read source s, in that order. */
#define MMVAE_FAN_MAX_BLOCKS 16
#define MMVAE_FAN_MAX_SRC 8
#define MMVAE_FAN_MAX_USES 4
typedef struct {
  const float* src[MMVAE_FAN_MAX_BLOCKS];
  float* dst[MMVAE_FAN_MAX_BLOCKS];
  int width[MMVAE_FAN_MAX_BLOCKS], ld_src[MMVAE_FAN_MAX_BLOCKS], ld_dst[MMVAE_FAN_MAX_BLOCKS];
  int n, B;
} mmvae_fan_blocks_t;
typedef struct {
  float* out[MMVAE_FAN_MAX_SRC];
  const float* g[MMVAE_FAN_MAX_SRC][MMVAE_FAN_MAX_USES];
  int ld[MMVAE_FAN_MAX_SRC][MMVAE_FAN_MAX_USES];
  int n_g[MMVAE_FAN_MAX_SRC], width[MMVAE_FAN_MAX_SRC];
  int n, B;
} mmvae_fan_sum_t;
int mmvae_rows_fan_fwd(const mmvae_fan_blocks_t* blocks, mmvae_stream_t stream);
int mmvae_rows_fan_bwd(const mmvae_fan_sum_t* sums, mmvae_stream_t stream);

/* MoE importance weights, models/mmvae_models.py:56-62:  lw[b] = sum_d [log N(z; mu_r, s_r) - log N(z; mu_o, s_o)]
 * with z and the source posterior detached; packed_* = (B,2D) [mu | sigma].  bwd: dpacked_r (B,2D) = g[b] * d lw. */
int mmvae_normal_logratio_fwd(const float* packed_r, const float* packed_o, const float* z, float* lw, int B, int D,
                              mmvae_stream_t stream);
int mmvae_normal_logratio_bwd(const float* packed_r, const float* z, const float* g, float* dpacked_r, int B, int D,
                              mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction losses (per-sample sums), ReconLoss.* in models/objectives.py
 * ---------------------------------------------------------------------------------------------- */
/* bce (objectives.py:392-406) on x_hat = clamp(sigmoid(logit),1e-6,1-1e-6) (decoders.py:96-97):
 *   row_loss[b] = sum_f -[t log xh + (1-t) log(1-xh)];  x_hat (B,F) given (already clamped).
 * target_rows: the target has that many rows and row b of x_hat is compared with target row b % target_rows -- a
 * K-sample decoder output (K*B rows) against the target repeated K times (BaseObjective.reshape_for_loss,
 * objectives.py:118-120) without materialising the repeat; also in _sigmoid_clamp_bwd and ce_over_time_fwd / _bwd. */
int mmvae_bce_rowsum_fwd(const float* x_hat, const float* target, float* row_loss, int B, int F, int target_rows,
                         mmvae_stream_t stream);
/* dlogit[b,f] = g[b] * (xh - t) * [1e-6 < xh < 1-1e-6]  (gradient through sigmoid+clamp+bce) */
int mmvae_bce_sigmoid_clamp_bwd(const float* x_hat, const float* target, const float* g_row, float* dlogit,
                                int B, int F, int target_rows, mmvae_stream_t stream);
/* true gradient wrt x_hat: dxh[b,f] = g[b] * (xh - t) / max(xh (1 - xh), 1e-12)  (torch's bce backward) */
int mmvae_bce_rowsum_bwd(const float* x_hat, const float* target, const float* g_row, float* dxhat, int B, int F,
                         mmvae_stream_t stream);
/* backward of y = clamp(sigmoid(l), 1e-6, 1-1e-6): dl = dy * y (1 - y) * [1e-6 < y < 1-1e-6] */
int mmvae_sigmoid_clamp_bwd(const float* dy, const float* y, float* dl, long n, mmvae_stream_t stream);
/* elementwise bce (B,F) for the ReconLoss.bce API */
int mmvae_bce_elem_fwd(const float* x_hat, const float* target, float* loss, long n, mmvae_stream_t stream);

/* Loss terms whose upstream gradient is a known constant (the ELBO is linear in them: seed = llik_scaling / B):
 * row sums AND the logit gradient seed * d(row)/d(logit) from one pass, so the loss backward costs no launch.
 * bce: x_hat = clamp(sigmoid(logit)), dlogit = seed (x_hat - t) where the clamp is inactive.
 * ce_over_time: MMVAE_ERR_UNSUPPORTED when T*V > 4096 or V > 256 (use fwd + bwd). */
int mmvae_bce_rowsum_seeded(const float* x_hat, const float* target, float* row_loss, float seed, float* dlogit,
                            int B, int F, mmvae_stream_t stream);
int mmvae_ce_over_time_seeded(const float* logits, const float* target, float* row_loss, float seed, float* dlogits,
                              int B, int T, int V, mmvae_stream_t stream);

/* Element-wise forms of the loss-plugin contract `ReconLoss.<name>(output, target, bs) -> (bs, -1)`
 * (models/objectives.py:389-509); the mixers' objective() uses the per-sample row sums instead.
 * lprob (:409-424): out[i] = -log p(target[i % target_n]) under Normal / Laplace(loc[i], scale) as a DOUBLE (the
 *   reference casts the fp32 element), NaN -> 0 (no gradient); scale <= 0: scale := loc.  bwd: g (n doubles).
 * optimal_sigma (:503-509): out[i] = detach(((t - x) / sigma)^2) + log_sigma + log sqrt(2 pi), ONE log_sigma =
 *   softclip(log sqrt(mean (t - x)^2), -6) per call; stats (4 floats) = {mean square, log_sigma, raw log sigma};
 *   ws: mmvae_optimal_sigma_ws_floats(1, n) floats.  bwd: g (n floats), gradient through log_sigma only.
 * pointwise: kind 0 = l1 |x - t| (:427-442), 1 = mse (x - t)^2 (:444-459); row sums (target row b % target_rows) with
 *   their backward, and elements: mmvae_pointwise_elem writes the loss (g == NULL) or g * d loss / d x. */
int mmvae_lprob_elem_fwd(const float* loc, const float* target, double* out, long n, long target_n, float scale,
                         int laplace, mmvae_stream_t stream);
int mmvae_lprob_elem_bwd(const float* loc, const float* target, const double* g, float* dloc, long n, long target_n,
                         float scale, int laplace, mmvae_stream_t stream);
int mmvae_optimal_sigma_elem_fwd(const float* loc, const float* target, float* out, float* stats, float* ws, long n,
                                 mmvae_stream_t stream);
int mmvae_optimal_sigma_elem_bwd(const float* loc, const float* target, const float* g, const float* stats, float* ws,
                                 float* dloc, long n, mmvae_stream_t stream);
int mmvae_pointwise_rowsum_fwd(const float* x, const float* target, float* row_loss, int B, int F, int target_rows,
                               int kind, mmvae_stream_t stream);
int mmvae_pointwise_rowsum_bwd(const float* x, const float* target, const float* g_row, float* dx, int B, int F,
                               int target_rows, int kind, mmvae_stream_t stream);
int mmvae_pointwise_elem(const float* x, const float* target, const float* g, float* out, long n, int kind,
                         mmvae_stream_t stream);

/* category_ce (objectives.py:486-500): softmax over TIME.  logits/target (B,T,V) ->
 *   loss (B,V) = -sum_t tgt * log_softmax_t(logits);  row_loss[b] = sum_v loss[b,v]  (either may be NULL) */
int mmvae_ce_over_time_fwd(const float* logits, const float* target, float* loss, float* row_loss, int B, int T,
                           int V, int target_rows, mmvae_stream_t stream);
/* dlogits[b,t,v] = g[b,v] * (softmax_t(logits)[t] * sum_t' tgt[t'] - tgt[t]);  g (B,V) or g_row (B) */
int mmvae_ce_over_time_bwd(const float* logits, const float* target, const float* g, const float* g_row,
                           float* dlogits, int B, int T, int V, int target_rows, mmvae_stream_t stream);

/* ReconLoss.lprob (models/objectives.py:409-424): row[b] = sum_f -log p(target[b,f]) under Normal (laplace = 0) or
 * Laplace(loc[b,f], scale); elements in fp32, the row sum in fp64 (the reference sums the elements as doubles), NaN
 * elements count 0 and get zero gradient.  scale <= 0: scale := loc (BaseObjective.recon_loss_fn, objectives.py:43-45,
 * when the modality has masks).  target_rows: the target has that many rows and row b of loc is compared with target
 * row b % target_rows -- the K-sample case, where the reference repeats the target K times
 * (BaseObjective.reshape_for_loss, objectives.py:118-120); target_rows == B otherwise.  lap_block_rows > 0: `laplace`
 * is a bit mask over consecutive blocks of that many rows (bit j: rows [j, j+1) * lap_block_rows are Laplace) -- one
 * launch for a decoder pass whose own / cross reconstructions carry different likelihood families (MoE,
 * models/mmvae_models.py:101-103 vs :115); 0: `laplace` is the flag for every row.
 * perm_c > 0: loc (and dloc) are stored (B, perm_c, F / perm_c) -- the NCHW output of a conv decoder -- while element j
 * of a target row pairs with loc[b, j % perm_c, j / perm_c]: Dec_SVHN returns its output permuted to (B,H,W,C) and
 * the loss reshapes (does not permute) the NCHW target to that shape (decoders.py:144, objectives.py:120); the kernel
 * indexes instead of materialising the permuted copy.
 * logit_grad != 0: loc holds y = sigmoid(logits) written by the producing layer's epilogue, and bwd emits the
 * gradient with respect to the LOGITS, dloc = g * d(-log p)/dy * y (1 - y) (no separate sigmoid backward pass). */
int mmvae_lprob_rowsum_fwd(const float* loc, const float* target, float* row_loss, int B, int F, int target_rows,
                           float scale, int laplace, int lap_block_rows, int perm_c, mmvae_stream_t stream);
int mmvae_lprob_rowsum_bwd(const float* loc, const float* target, const float* g_row, float* dloc, int B, int F,
                           int target_rows, float scale, int laplace, int lap_block_rows, int perm_c, int logit_grad,
                           mmvae_stream_t stream);
/* ReconLoss.optimal_sigma (models/objectives.py:503-509) + utils.softclip (utils.py:66-69): one log sigma per call
 * from the mean squared error over all B*F elements; row[b] = sum_f ((t-x)/sigma)^2 + F (log sigma + log sqrt(2 pi)).
 * stats (3 floats, written by fwd, read by bwd) = {mean square, log sigma, unclipped log sigma}.  Gradient flows only
 * through log sigma (the squared term is detached in the reference). */
size_t mmvae_optimal_sigma_ws_floats(int B, int F);
int mmvae_optimal_sigma_fwd(const float* loc, const float* target, float* row_loss, float* stats, float* ws, int B,
                            int F, mmvae_stream_t stream);
int mmvae_optimal_sigma_bwd(const float* loc, const float* target, const float* g_row, const float* stats,
                            float* dloc, int B, int F, mmvae_stream_t stream);

/* out[k] = sum_n W[k,n] * sum_b V[n,b]   (ELBO assembly: BaseObjective.elbo objectives.py:54-67,
 * MoPOE.objective mmvae_models.py:315-319).  W is passed by value (<= 4 x 32 host floats). */
int mmvae_lincomb_rows_fwd(const float* V, const float* W_host, float* out, int n_rows, int B, int n_out,
                           mmvae_stream_t stream);
int mmvae_lincomb_rows_bwd(const float* gout, const float* W_host, float* dV, int n_rows, int B, int n_out,
                           mmvae_stream_t stream);

/* The same with the rows in separate tensors (p[n] -> B floats) and one upstream-gradient scalar per output
 * (g[k] device pointer or NULL = that output does not take part in backward); d rows are written to drows->p[n].
 * d_unit (may be NULL): fwd also writes the row gradients of output 0 for a unit upstream gradient (the constants
 * W[0][n]) -- a training step seeds loss.backward() with ones, so its backward needs no launch at all. */
typedef struct {
  const float* p[32];
} mmvae_rowptrs_t;
typedef struct {
  const float* g[4];
} mmvae_gptrs_t;
int mmvae_lincomb_rowptrs_fwd(const mmvae_rowptrs_t* rows, const float* W_host, float* out,
                              const mmvae_rowptrs_t* d_unit, int n_rows, int B, int n_out, mmvae_stream_t stream);
int mmvae_lincomb_rowptrs_bwd(const mmvae_gptrs_t* gout, const float* W_host, const mmvae_rowptrs_t* drows,
                              int n_rows, int B, int n_out, mmvae_stream_t stream);

/* MoE ELBO (models/mmvae_models.py:61-77): wc = exp(lw) * r;  loss = (sum_n W_n rowsum_n + n_nz beta sum kld) / M
 * where n_nz counts the rows whose weighted sum is not exactly 0 (the reference's `lp.sum() != 0` filter and the
 * broadcast in BaseObjective.elbo).  out[0] = loss, out[1] = n_nz (kept for the backward). */
int mmvae_expmul_fwd(const float* lw, const float* r, float* out, int n, mmvae_stream_t stream);
int mmvae_expmul_bwd(const float* lw, const float* r, const float* g, float* dlw, float* dr, int n,
                     mmvae_stream_t stream);
int mmvae_moe_elbo_fwd(const float* rows, const float* W_host, const float* kld, float* out, int n_rows, int M, int B,
                       float beta, mmvae_stream_t stream);
int mmvae_moe_elbo_bwd(const float* g, const float* out, const float* W_host, float* drows, float* dkld, int n_rows,
                       int M, int B, float beta, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ResNet-50 image tower (`encoder: CNN`, models/encoders.py:86-127: torchvision resnet50 -> SiLU -> heads).  Inside the
 * tower activations are NHWC (rows = B*H*W, C) matrices; layers emit pre-activations, `in_act` (NONE / RELU) is applied
 * while reading.
 * ---------------------------------------------------------------------------------------------- */
/* AdaptiveAvgPool2d(1) on act(x): (B, HW, C) -> (B, C) */
int mmvae_avgpool_fwd(const float* x, float* y, int B, int HW, int C, int in_act, mmvae_stream_t stream);
int mmvae_avgpool_bwd(const float* dy, const float* x, float* dx, int B, int HW, int C, int in_act,
                      mmvae_stream_t stream);

/* ---- fused convolution + BatchNorm engine of the bottleneck stack (csrc/rconv.hip; replaces the reference's
 * torchvision Bottleneck.forward = conv1x1 -> bn -> relu -> conv3x3 -> bn -> relu -> conv1x1 -> bn (+ shortcut) and its
 * autograd backward, models/encoders.py:108).  Weights are channels-last: (Cout, T = k*k taps, Cin) in memory behind the
 * (Cout, Cin, k, k) parameter view.  A convolution's output stays RAW (pre-BatchNorm); its consumers apply
 * bn(y) = fma(y - mean, sc, beta), sc = gamma * rstd, and the ReLU while staging their LDS tiles.  `pre`: how the input is
 * consumed -- 0 as it is, 1 relu(x), 2 relu(bn(x)) with (xmean, xsc, xbeta).  All channel counts % 64 == 0. */
/* fwd (T, B*Ho*Wo) and bwd (T, B*H*W) int32 tables of a K x K / stride S / padding P convolution (-1 = no source) */
int mmvae_rc_tables(int* fwd, int* bwd, int B, int H, int W, int K, int S, int P, mmvae_stream_t stream);
/* rows per statistics partial the kernels use for an (M rows, N columns) output (64).  With R = ceil(M / 64) row tiles
 * a `part` buffer holds (R + ceil(R / 16)) * N * 2 floats and a `counter` buffer (N / 64) * (1 + ceil(R / 16)) tickets
 * (zero once; the kernels re-arm them): more than 16 row tiles elect their finalizer in two levels. */
int mmvae_rc_row_tile(int M, int N);
/* forward / data-gradient GEMMs with few output tiles split their reduction (K channels x T taps) over workgroups:
 * ws = mmvae_rc_conv_ws_floats(M, N, K, T) floats and (ceil(M / 64) + 4) * (N / 64) tile tickets (zero once) for an (M, N)
 * output (forward: N = Cout, K = Cin; data gradient: M = Min, N = Cin, K = Cout) */
int mmvae_rc_conv_splits(int M, int N, int K, int T);
size_t mmvae_rc_conv_ws_floats(int M, int N, int K, int T);
/* a BatchNorm whose output gradient a kernel produces: the kernel sums (G, G xhat) over the rows and its last workgroup
 * writes dgamma / dbeta ((+)= per acc) and pqr (3, C) with  dL/dY = G p + Y q + r  (torch.nn.BatchNorm2d backward,
 * training mode; eval: p = gamma rstd, q = r = 0) */
typedef struct {
  const float* Y;
  const float* mean;
  const float* rstd;
  const float* gamma;
  float* pqr;
  float* dgamma;
  float* dbeta;
  float* part;
  unsigned* counter;
  int acc;
  int eval;
} mmvae_rc_stat_t;
/* pixel geometry of a convolution: (B, H, W) input pixels -> (B, Ho, Wo) output pixels, KW columns of taps (T / KW rows),
 * stride S, padding P; H, W < 16384 */
typedef struct {
  int H, W, Ho, Wo, KW, S, P;
} mmvae_rc_geom_t;
/* forward: y (M, Cout) = conv(pre(x)); with part != NULL also the BatchNorm that follows: batch mean / rstd /
 * sc = gamma rstd out, running statistics moved (eval != 0: mean / rstd / sc from the running statistics, nothing moved) */
typedef struct {
  const float* x;
  const float* w;
  const float* xmean;
  const float* xsc;
  const float* xbeta;
  float* y;
  float* ws;
  unsigned* tile_ticket;
  int M, Cin, Cout, T, pre;
  mmvae_rc_geom_t g;
  const float* gamma;
  const float* beta;
  float* run_mean;
  float* run_var;
  float* mean;
  float* rstd;
  float* sc;
  float* part;
  unsigned* counter;
  float eps, momentum;
  int eval;
} mmvae_rc_fwd_t;
/* data gradient: out (Min, Cin) = mask * (conv_transpose(dY) + add), dY = G p + Y q + r per output channel (pqr NULL:
 * dY = G); add: (Min, Cin), or -- with add_tbl (Min) -- row add_tbl[m] of it (-1: none); mask: 0 none, 1 mY > 0,
 * 2 bn(mY) > 0 with (mmean, msc, mbeta); nstat BatchNorms (same rows / channels as out) get their backward
 * statistics from the epilogue */
typedef struct {
  const float* G;
  const float* Y;
  const float* pqr;
  const float* w;
  const float* add;
  const int* add_tbl;
  int mask;
  const float* mY;
  const float* mmean;
  const float* msc;
  const float* mbeta;
  float* out;
  float* ws;
  unsigned* tile_ticket;
  const int* row_map; /* stride 2, even H and W: the Min input pixels ordered by the parity class (ih % 2, iw % 2), classes
                         (0,0) (0,1) (1,0) (1,1), raster order inside a class -- a tile then only runs its class's taps
                         (9 / 4 instead of 9 passes per pixel for 3 x 3); statistics buffers sized for 4 * ceil(Min / 256)
                         row tiles.  NULL: raster order */
  int M, Min, Cin, Cout, T, nstat;
  mmvae_rc_geom_t g;
  mmvae_rc_stat_t st[2];
} mmvae_rc_dgrad_t;
/* weight gradient: dw (Cout, T, Cin) (+)= dY^T pre(x) per tap; tbl: the forward table of mmvae_rc_tables (NULL: 1x1,
 * stride 1); ws / counter: mmvae_rc_wgrad_ws_floats / _tickets */
typedef struct {
  const float* G;
  const float* Y;
  const float* pqr;
  const float* x;
  const float* xmean;
  const float* xsc;
  const float* xbeta;
  const int* tbl;
  float* dw;
  float* ws;
  unsigned* counter;
  int M, Cin, Cout, T, pre, accumulate;
} mmvae_rc_wgrad_t;
/* up to MMVAE_RC_MAX_JOBS independent jobs in ONE launch (a block's first convolution + projection shortcut, the
 * data gradients that only need one incoming gradient, batches of weight gradients beside the data-gradient chain);
 * kind: 0 forward (f), 1 data gradient (d), 2 weight gradient (w) */
#define MMVAE_RC_MAX_JOBS 8
typedef struct {
  int kind;
  mmvae_rc_fwd_t f;
  mmvae_rc_dgrad_t d;
  mmvae_rc_wgrad_t w;
} mmvae_rc_job_t;
int mmvae_rc_launch(const mmvae_rc_job_t* jobs, int n, mmvae_stream_t stream);
/* the same statistics for a gradient produced elsewhere (pooling backward) */
int mmvae_rc_bn_bwd_stats(const float* G, const mmvae_rc_stat_t* st, int M, int C, mmvae_stream_t stream);
/* ... and for the gradient of AdaptiveAvgPool2d(1) on relu(x) (torchvision resnet50.avgpool behind the last bottleneck),
 * made in the same launch: G (B*HW, C) = dy[b, c] / HW * (x > 0) */
int mmvae_rc_pool_bwd_stats(const float* dy, const float* x, float* G, const mmvae_rc_stat_t* st, int B, int HW, int C,
                            mmvae_stream_t stream);
int mmvae_rc_wgrad_splits(int M, int Cin, int Cout, int T);
size_t mmvae_rc_wgrad_ws_floats(int M, int Cin, int Cout, int T);
size_t mmvae_rc_wgrad_tickets(int Cin, int Cout, int T);
/* the stem (torchvision resnet50.conv1 / bn1 / relu / maxpool): y (M, Cout) = conv(img) of the NCHW image batch
 * (B, Cimg, H, W) with the channels-last weight (Cout, T, Cimg), BatchNorm statistics as in the forward jobs; max pooling
 * (3, 2, 1) of relu(bn(y)) with the index of each window's first maximum; its backward made in the launch that sums the
 * BatchNorm's backward statistics, G (B*H*W, C) = (bn(Y) > 0) * scattered dy (st->Y = y, st->mean with sc / beta
 * normalise); and the weight gradient, dY = G p + Y q + r, tbl (2, M): per output pixel b Cimg H W + h0 W + w0 and
 * (h0 + 16384) << 16 | (w0 + 16384) of its window origin (h0, w0) = (oh S - P, ow S - P). */
int mmvae_rc_stem_fwd(const float* img, const float* w, float* y, int M, int Cimg, int Cout, int T,
                      const mmvae_rc_geom_t* g, const float* gamma, const float* beta, float* run_mean, float* run_var,
                      float* mean, float* rstd, float* sc, float* part, unsigned* counter, float eps, float momentum,
                      int eval, mmvae_stream_t stream);
int mmvae_rc_maxpool_fwd(const float* y, const float* mean, const float* sc, const float* beta, float* out, int* idx, int B,
                         int H, int W, int C, mmvae_stream_t stream);
int mmvae_rc_maxpool_bwd_stats(const float* dy, const int* idx, float* G, const float* sc, const float* beta,
                               const mmvae_rc_stat_t* st, int B, int H, int W, int C, mmvae_stream_t stream);
int mmvae_rc_stem_wgrad_splits(int M, int Cimg, int Cout, int T);
size_t mmvae_rc_stem_wgrad_ws_floats(int M, int Cimg, int Cout, int T);
/* ws: mmvae_rc_stem_wgrad_ws_floats floats; counter: (Cout / 64) * ceil(T Cimg / 64) tickets (zero once) */
int mmvae_rc_stem_wgrad(const float* G, const float* Y, const float* pqr, const float* img, const int* tbl, float* dw,
                        float* ws, unsigned* counter, int M, int Cimg, int Cout, int T, const mmvae_rc_geom_t* g,
                        int accumulate, mmvae_stream_t stream);
/* end of a bottleneck: out = bn3(Y3) + (bn_d(R) when mr != NULL, else relu?(R)) */
int mmvae_rc_blockout(const float* Y3, const float* m3, const float* sc3, const float* b3, const float* R,
                      const float* mr, const float* scr, const float* br, int res_relu, float* out, long rows, int C,
                      mmvae_stream_t stream);
/* out = bn(Y) in the engine's own arithmetic (diagnostics, ReLU-mask export) */
int mmvae_rc_bn_apply(const float* Y, const float* mean, const float* sc, const float* beta, float* out, long rows, int C,
                      mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MoE with K samples per posterior + the DReG objective: the shipped configs/config_mnistsvhn.yml (mixing moe,
 * obj dreg, K 30, prior laplace).  Replaces MOE.forward's `q_m.rsample([K])` (models/mmvae_models.py:96-100),
 * the non-elbo branch of MOE.objective (:63-78) and MultimodalObjective.dreg / _m_dreg_looser
 * (models/objectives.py:361-387).
 *   packed[m] (B, 2D) = [mu_m | scale_m] head outputs (softmax + 1e-6 already applied); laplace[m] != 0: q_m is a
 *   Laplace (the config's `prior` key also selects the posterior family, models/trainer.py:104), else a Normal.
 *   eps[m] (K,B,D): standard Normal / Laplace variates (mmvae_randn / mmvae_rand_laplace);  z[m] (K,B,D) out.
 *   lat (M,K,B)   = sum_d log N(z_r[k,b]; 0, softmax(theta) D) - beta * log-mean-exp_m sum_d log q_m(z_r[k,b])
 *                   (beta = 1 for dreg, objectives.py:372; the objective's beta for iwae, objectives.py:356)
 *   pi  (M,K,B,M) = softmax_m of those row sums (saved for the backward pass).
 * bwd: dlat (M,K,B), dz[m] (K,B,D) or NULL -> dpacked[m] (B,2D) [written], dtheta_rows (B,D) [row b = sample b's
 *   contribution to d theta; the caller folds the rows; NULL = not wanted].  D <= 256, 2 <= M <= 4.
 * ---------------------------------------------------------------------------------------------- */
#define MMVAE_MOE_MAX_MODS 4
typedef struct {
  const float* packed[MMVAE_MOE_MAX_MODS];
  const float* eps[MMVAE_MOE_MAX_MODS];
  float* z[MMVAE_MOE_MAX_MODS];
  int laplace[MMVAE_MOE_MAX_MODS];
} mmvae_moe_k_args;
typedef struct {
  const float* packed[MMVAE_MOE_MAX_MODS];
  const float* eps[MMVAE_MOE_MAX_MODS];
  const float* z[MMVAE_MOE_MAX_MODS];
  const float* dz[MMVAE_MOE_MAX_MODS];
  float* dpacked[MMVAE_MOE_MAX_MODS];
  int laplace[MMVAE_MOE_MAX_MODS];
} mmvae_moe_k_bwd_args;
int mmvae_moe_ksample_fwd(const mmvae_moe_k_args* a, const float* theta, float* lat, float* pi, int M, int K, int B,
                          int D, float beta, mmvae_stream_t stream);
int mmvae_moe_ksample_bwd(const mmvae_moe_k_bwd_args* a, const float* theta, const float* dlat, const float* pi,
                          float* dtheta_rows, int M, int K, int B, int D, float beta, mmvae_stream_t stream);
/* DReG loss (objectives.py:375-387).  own[r] / cross[r] (K*B): POSITIVE per-sample reconstruction sums of modality r
 * decoded from its own / the other modality's samples (mmvae_lprob_rowsum_fwd); lam[r] = llik_scaling.
 *   lw[r,k] = sum_b lat[r,k,b] - lam_r sum_b (own_r + cross_r)[k,b]   (fp64 sums);  w = softmax_k lw, detached;
 *   out (doubles, 1 + 2 M K + 2 M K): [0] loss = -(1/M) sum_rk w lw | lw (M,K) | w (M,K) | lpx (M,2,K) [own, cross]
 * bwd: g = d loss (device double) -> dlat (M,K,B) = -g w / M, d own = d cross = g w lam / M. */
typedef struct {
  const float* own[MMVAE_MOE_MAX_MODS];
  const float* cross[MMVAE_MOE_MAX_MODS];
  float lam[MMVAE_MOE_MAX_MODS];
} mmvae_dreg_rows;
typedef struct {
  float* own[MMVAE_MOE_MAX_MODS];
  float* cross[MMVAE_MOE_MAX_MODS];
  float lam[MMVAE_MOE_MAX_MODS];
} mmvae_dreg_rows_grad;
int mmvae_dreg_loss_fwd(const float* lat, const mmvae_dreg_rows* rows, double* out, int M, int K, int B,
                        mmvae_stream_t stream);
int mmvae_dreg_loss_bwd(const double* out, const double* g, const mmvae_dreg_rows_grad* rows, float* dlat, int M, int K,
                        int B, mmvae_stream_t stream);
/* IWAE loss (MultimodalObjective.iwae, objectives.py:342-359; the K > 1 AND B > 1 case is a defined extension, see
 * models/objectives.py of this package).  Same inputs as the DReG loss, per SAMPLE instead of summed over the batch:
 *   lw[r,k,b] = lat[r,k,b] - lam_r (own_r + cross_r)[k,b];  loss = -sum_b (logsumexp_{r,k} lw[.,.,b] - log(M K)), fp64
 *   out (doubles, mmvae_iwae_loss_out_doubles): [0] loss | lse (B) | lpx (M,2,K*B) [own, cross]
 * bwd: g = d loss (device double) -> dlat = -g softmax_{rk}(lw), d own = d cross = g softmax lam_r. */
size_t mmvae_iwae_loss_out_doubles(int M, int K, int B);
int mmvae_iwae_loss_fwd(const float* lat, const mmvae_dreg_rows* rows, double* out, int M, int K, int B,
                        mmvae_stream_t stream);
int mmvae_iwae_loss_bwd(const float* lat, const double* out, const double* g, const mmvae_dreg_rows_grad* rows,
                        float* dlat, int M, int K, int B, mmvae_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Enc_TxtRNN (models/encoders.py:840-869): Embedding -> bidirectional one-layer GRU(512) -> output[-1] -> sum of the
 * directions -> Linear -> chunk -> softmax + eta.  A DEFINED path, parity unpinned against the reference (its forward
 * crashes on its own batch format); oracle/mmvae_oracle.py: enc_txt_rnn, pinned to torch.nn.GRU.
 *   token_ids: onehot (B,T,V) -> ids (T,B) int32 [argmax; all-zero padding rows = token 0] + canonical one-hot (T*B,V).
 *   pt (3H,V) = W_ih E^T (the embedding folded into the input projection), bih / whh (3H,H) / bhh: torch's GRU
 *   parameters (gate order r, z, n).  forward: hs (T+1,B,H) with hs[0] = 0 given, hs[t+1] = h after step t (one launch
 *   per step: the (B,H)x(H,3H) product on 16x16x4 fp32 MFMA tiles with the gates in the epilogue);
 *   saved (4,T,B,H) = r | z | n | q (q = W_hn h + b_hn).  H % 64 == 0.
 *   backward: dh (B,H) = d loss / d h_T in (destroyed) -> dgx, dgh (T,B,3H): the input-side / hidden-side gate
 *   derivatives of every step (d Pt^T, d b_ih = reductions of dgx; d W_hh, d b_hh = reductions of dgh over hs[0:T]).
 *   cell0: the reverse direction at the last position = ONE cell step from h = 0 on the last token; out = hfwd + h;
 *   saved (3,B,H) = r | z | n;  bwd: dout (B,H) -> dgx, dgh (B,3H).
 * ---------------------------------------------------------------------------------------------- */
int mmvae_gru_token_ids(const float* onehot, int* ids, float* onehot_tb, int B, int T, int V, mmvae_stream_t stream);
int mmvae_gru_forward(const float* pt, const float* bih, const float* whh, const float* bhh, const int* ids, float* hs,
                      float* saved, int T, int B, int H, int V, mmvae_stream_t stream);
int mmvae_gru_backward(float* dh, const float* whh, const float* hs, const float* saved, float* dgx, float* dgh, int T,
                       int B, int H, mmvae_stream_t stream);
int mmvae_gru_cell0_fwd(const float* pt, const float* bih, const float* bhh, const int* ids, const float* hfwd,
                        float* out, float* saved, int B, int H, int V, mmvae_stream_t stream);
int mmvae_gru_cell0_bwd(const float* dout, const float* saved, const float* bhh, float* dgx, float* dgh, int B, int H,
                        mmvae_stream_t stream);

/* `prior: laplace` with the elbo objective: KL(Laplace(mu, s) || N(0,1)) row sums (torch's _kl_laplace_normal reached
 * through utils.kl_divergence, utils.py:399-402; models/mmvae_models.py:45) and the importance ratio of :56-62 under
 * Laplace posteriors (gradient into packed_r only, as mmvae_normal_logratio_*). */
int mmvae_kl_laplace_normal_fwd(const float* packed, float* kl, int B, int D, mmvae_stream_t stream);
int mmvae_kl_laplace_normal_bwd(const float* packed, const float* g, float* dpacked, int B, int D,
                                mmvae_stream_t stream);
int mmvae_laplace_logratio_fwd(const float* packed_r, const float* packed_o, const float* z, float* lw, int B, int D,
                               mmvae_stream_t stream);
int mmvae_laplace_logratio_bwd(const float* packed_r, const float* z, const float* g, float* dpacked_r, int B, int D,
                               mmvae_stream_t stream);
/* standard-Laplace variates e = -sign(u) log1p(-|u|), u ~ U(-1,1) (torch.distributions.Laplace.rsample: z = loc +
 * scale e); generator state as mmvae_randn */
int mmvae_rand_laplace(float* out, long n, uint32_t* state, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Text towers (Enc_TxtTransformer / Dec_TxtTransformer, models/encoders.py:790-837, decoders.py:668-723)
 * ---------------------------------------------------------------------------------------------- */
/* Embedding(one-hot.long()) + PositionalEncoding quirk (models/nn_modules.py:430-438, encoders.py:833-835).
 *   onehot (B,T,V) 0/1 floats, emb (V,2) [only rows 0/1 are read], pe (>= max(B,T), 2) = the module's
 *   sin/cos buffer (host-computed once), x (T*B, 2V) = the (nframes,bs,-1) view.
 *   mode 0 (B != T, B != 1): x[t,b,v,e] = emb[oh[b,t,v],e] + pe[b,e]
 *   mode 1 (B == T or B == 1): memory (B,T,2V) with pe[t] (pe[0] if B == 1), relabelled as (T,B,2V).
 *   B0 (round 5): onehot has B0 rows, B = R * B0 output rows, row b reads onehot row and pe row b % B0 (mode 0 only
 *   when B0 < B): R passes over the same batch as one call (POE's per-subset passes, models/mmvae_models.py:159-187). */
int mmvae_embed_pe_fwd(const float* onehot, const float* emb, const float* pe, float* x, int B, int T, int V,
                       int mode, int B0, const mmvae_dropout_t* drop, mmvae_stream_t stream);
int mmvae_embed_pe_bwd(const float* onehot, const float* dx, float* demb, float* ws, int B, int T, int V, int mode,
                       int B0, int accumulate, const mmvae_dropout_t* drop, mmvae_stream_t stream);
size_t mmvae_embed_ws_floats(int B, int T, int V);

/* ---- One Transformer layer of the text towers per launch (csrc/txtlayer.hip) --------------------------------
 * torch.nn.TransformerEncoderLayer / TransformerDecoderLayer (post-norm, gelu, dropout p) as the reference builds
 * them: models/encoders.py:806-812 (Enc_TxtTransformer, d = 2V = 54, ff 128, 2 heads) and models/decoders.py:686-692
 * (Dec_TxtTransformer, d = n_latents, memory length 1).  One workgroup per sequence, L <= 32 tokens; activations are
 * (L, N, D) time-major like the rest of the text path.  Shapes outside mmvae_txt_layer_supported() use the op-by-op
 * entry points below (same arithmetic, same dropout masks).
 *   fwd writes y and the tensors backward needs; bwd runs the whole data-gradient chain and leaves the output
 *   gradient of every GEMM in HBM (d_*), from which the weight gradients are mmvae_linear_bwd_weight calls over the
 *   (L*N)-row tensors:  in_proj (d_qkv, x) | out_proj (d_a, ao) | linear1 (d_h1, x1 or x2) | linear2 (d_f, g)
 *   decoder: cross out_proj (d_ca, vb) | cross value projection (d_v (N,D), mem).
 *   lnws (N, n_norms, 2, D): per-sequence partials of the LayerNorm gamma / beta gradients (norm order 1, [2,] last). */
typedef struct {
  const float *in_w, *in_b;     /* self-attention in_proj (3D, D), (3D) */
  const float *out_w, *out_b;   /* self-attention out_proj (D, D), (D) */
  const float *l1_w, *l1_b;     /* linear1 (FF, D), (FF) */
  const float *l2_w, *l2_b;     /* linear2 (D, FF), (D) */
  const float *n1_g, *n1_b, *n2_g, *n2_b, *n3_g, *n3_b; /* LayerNorm weight / bias; n3 decoder only */
  const float *x_in_w, *x_in_b; /* decoder: VALUE rows of the cross-attention in_proj (D, D), (D) */
  const float *x_out_w, *x_out_b; /* decoder: cross-attention out_proj */
} mmvae_txt_layer_w_t;
typedef struct {
  float *qkv;                   /* (L,N,3D) */
  float *probs;                 /* (N,H,L,L) normalised attention weights before dropout, or NULL */
  float *ao;                    /* (L,N,D) attention output (input of out_proj) */
  float *xhat1, *rstd1, *x1;    /* LayerNorm1: (L,N,D), (L,N), output (L,N,D) */
  float *vproj, *vb;            /* decoder: value projection (N,D); its dropout-masked broadcast (L,N,D) */
  float *xhat2, *rstd2, *x2;    /* decoder LayerNorm2 */
  float *h1, *g;                /* linear1 output (L,N,FF); dropout(gelu(h1)) (L,N,FF) */
  float *xhatf, *rstdf;         /* last LayerNorm */
} mmvae_txt_layer_saved_t;
typedef struct {
  float *d_f, *d_h1, *d_ca, *d_v, *d_a, *d_qkv, *lnws;
} mmvae_txt_layer_grads_t;
typedef struct {
  mmvae_dropout_t attn, drop1, xattn, drop2, ffn, drop3; /* encoder: attn, drop1, ffn, drop2 */
} mmvae_txt_layer_drop_t;
int mmvae_txt_layer_supported(int L, int D, int FF, int NH, int dec);
/* Round 4: two kernel families serve these entry points -- one workgroup per sequence (csrc/txtlayer.hip) and one WAVE
 * per sequence with the activations chained through registers (csrc/txtwave.hip; forward always, backward from 384
 * sequences).  Same arithmetic, saved tensors and masks; this sets the choice (-1 keeps a value): forward on the wave
 * kernels, backward on them from bwd_min_n sequences (d > 32 layers) / bwd_min_n_dec (d <= 32 decoder layers).
 * Defaults from MMVAE_TXT_WAVE / MMVAE_TXT_WAVE_BWD_MIN_N[_DEC]. */
int mmvae_txt_layer_plan(int fwd_wave, int bwd_min_n, int bwd_min_n_dec);
size_t mmvae_txt_layer_lnws_floats(int N, int D, int dec);
/* time_mean != 0: y / dy are (N, D), the mean of the layer output over the L frames and its gradient -- the pooling
 * `x.mean(0)` that follows the last encoder layer (models/encoders.py:552), folded into the layer's launch */
/* head_w (HN, D), head_b (HN) non-NULL (needs time_mean): additionally heads (N, HN) = y head_w^T + head_b, the
 * packed posterior heads on the pooled feature (VaeComponent.process_output, models/encoders.py:49-54); their
 * backward is an ordinary mmvae_linear_bwd on (heads gradient, y) */
int mmvae_txt_layer_fwd(const float* x, const uint8_t* valid, const float* mem, float* y,
                        const mmvae_txt_layer_w_t* w, const mmvae_txt_layer_saved_t* saved,
                        const mmvae_txt_layer_drop_t* drop, int L, int N, int D, int FF, int NH, int dec,
                        int time_mean, const float* head_w, const float* head_b, float* heads, int HN,
                        mmvae_stream_t stream);
int mmvae_txt_layer_bwd(const float* dy, const uint8_t* valid, float* dx, float* dmem,
                        const mmvae_txt_layer_w_t* w, const mmvae_txt_layer_saved_t* saved,
                        const mmvae_txt_layer_grads_t* grads, const mmvae_txt_layer_drop_t* drop, int L, int N, int D,
                        int FF, int NH, int dec, int time_mean, mmvae_stream_t stream);

/* Input expansion on the device (the step in front of the path): bit-identical to the reference's
 * host-side `torch.tensor(uint8) / 255` (models/datasets.py:251-254) and `one_hot_encode` + `lengths_to_mask`
 * (utils.py:414-421, :239; models/datasets.py:272-281).  tokens (B,T) int32, -1 = character outside the alphabet (zero
 * row); lengths (B); onehot (B,T,V) fp32; mask (B,T) bytes, may be NULL. */
int mmvae_expand_image_u8(const uint8_t* src, float* dst, long n, mmvae_stream_t stream);
int mmvae_expand_text_tokens(const int32_t* tokens, const int32_t* lengths, float* onehot, uint8_t* mask, int B, int T,
                             int V, mmvae_stream_t stream);
/* The host-to-device transfer of the NEXT batch as a node inside the captured step: ring_dev = device array of n_ring
 * pinned (device-visible, 16-byte aligned) host batches of `bytes` bytes; ctr_dev = two device words {count, ticket}
 * (zeroed once); the launch copies bytes [offset, offset + bytes) of ring[count % n_ring] into the same range of
 * `staging` and, when `advance` is set, advances count itself (a batch may be pulled in several parts: the last one
 * advances). */
int mmvae_input_ring_pull(const void* const* ring_dev, int n_ring, unsigned* ctr_dev, void* staging, size_t offset,
                          size_t bytes, int advance, mmvae_stream_t stream);

/* The input step as a native pipe: a packed pinned host batch (every tensor a 16-byte aligned slice of one buffer) is
 * copied with ONE H2D transfer on the pipe's own copy stream into `staging` (device, `bytes` long, owned by the
 * caller), under the step that is running; mmvae_input_pipe_commit then, on `stream`: waits for that copy, expands
 * every modality into the step's static inputs (the two functions above), marks the staging buffer consumed and --
 * `next_host_packed` != NULL -- starts the copy of the following batch behind that.  One call per step.
 * Replaces, on the caller side of the path, the reference's per-step DataLoader -> device transfer of fp32 images and
 * one-hot text (models/dataloader.py:120-126; tensors built by models/datasets.py:251-254, :272-281). */
#define MMVAE_INPUT_MAX_MODS 8
#define MMVAE_INPUT_IMAGE_U8 0
#define MMVAE_INPUT_TEXT_TOKENS 1
typedef struct mmvae_input_pipe mmvae_input_pipe_t;
typedef struct {
  int kind;         /* MMVAE_INPUT_IMAGE_U8 | MMVAE_INPUT_TEXT_TOKENS */
  size_t src_off;   /* byte offset in the packed batch: uint8 pixels | (B,T) int32 tokens */
  size_t len_off;   /* text: byte offset of the (B) int32 lengths */
  float* dst;       /* fp32 image (n values) | one-hot (B,T,V) */
  uint8_t* mask;    /* text: (B,T) bytes or NULL */
  long n;           /* image: number of pixel values */
  int B, T, V;      /* text */
} mmvae_input_mod_t;
int mmvae_input_pipe_create(mmvae_input_pipe_t** out, void* staging, size_t bytes);
int mmvae_input_pipe_destroy(mmvae_input_pipe_t* pipe);
int mmvae_input_pipe_prefetch(mmvae_input_pipe_t* pipe, const void* host_packed);
int mmvae_input_pipe_commit(mmvae_input_pipe_t* pipe, const mmvae_input_mod_t* mods, int n_mods,
                            const void* next_host_packed, mmvae_stream_t stream);

/* y[t,b,:] = dropout(x[t,b,:] + pe[t,:]) -- Enc_Transformer / Dec_Transformer positional encoding
 * (models/encoders.py:721-723, models/decoders.py:607-608; PositionalEncoding try-branch nn_modules.py:430-438).
 * x may be NULL (time queries PE(zeros)).  Backward: mmvae_dropout_act_bwd with MMVAE_ACT_NONE. */
int mmvae_add_pe_dropout_fwd(const float* x, const float* pe, float* y, int T, int B, int D,
                             const mmvae_dropout_t* drop, mmvae_stream_t stream);

/* Scaled-dot-product attention with key padding mask for L,S <= 64 (nn.MultiheadAttention core).
 *   q (L*N, ldq) rows r = l*N+n, head h at columns [h*hd,(h+1)*hd); k, v (S*N, ld) likewise.
 *   kpm (N,S) bytes, 1 = ignore key (may be NULL); with mask_is_valid != 0 the bytes are the batch's validity
 *   mask instead (1 = real token).  out (L*N, E=H*hd).  probs (N,H,L,S) saved for bwd: the normalised weights before
 *   dropout, a weight that the dropout removed stored NEGATED (sign bit = the mask: mmvae_attn_bwd hashes nothing). */
int mmvae_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* kpm, float* out, float* probs,
                   int L, int S, int N, int H, int hd, long ldq, long ldk, long ldv, int mask_is_valid,
                   const mmvae_dropout_t* drop, mmvae_stream_t stream);
int mmvae_attn_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                   float* dq, float* dk, float* dv, int L, int S, int N, int H, int hd, long ldq, long ldk,
                   long ldv, const mmvae_dropout_t* drop, mmvae_stream_t stream);
/* runtime switch between the two MFMA backward kernels of mmvae_attn_bwd for 32 < L, S <= 128: 0 (default) = the LDS-tile form,
 * 1 = the register form (csrc/text.hip: attn_t_bwd_kernel, needs 16-byte aligned head slices; measured level in the step);
 * returns the previous setting.  MMVAE_ATTN_T_BWD=1 in the environment starts with 1.  (nn.MultiheadAttention backward: reference models/encoders.py:706-716) */
int mmvae_attn_t_bwd_set(int on);

/* y = LayerNorm(x + r) * gamma + beta (eps 1e-5); r may be NULL, same-shape (r_rows = 0) or broadcast over time
 * (r_rows = N: row index = row % N).  xhat (rows,d) and rstd (rows) are saved for the backward.
 * With dropout: y = LN(dropout(x) + r).
 * bwd: dsum = grad wrt the sum (= grad of r); with dropout dx_drop = dsum * mask is the grad of x;
 * dgamma/dbeta (+)= through ws (mmvae_layernorm_ws_floats). */
int mmvae_layernorm_residual_fwd(const float* x, const float* r, const float* gamma, const float* beta, float* y,
                                 float* xhat, float* rstd, int rows, int d, int r_rows, const mmvae_dropout_t* drop,
                                 mmvae_stream_t stream);
int mmvae_layernorm_residual_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                                 float* dsum, float* dx_drop, float* dgamma, float* dbeta, float* ws, int rows, int d,
                                 int accumulate, const mmvae_dropout_t* drop, mmvae_stream_t stream);
size_t mmvae_layernorm_ws_floats(int rows, int d);

/* mean over the leading (time) axis: x (L,N,d) -> y (N,d)  (encoders.py:836); bwd: dx = dy / L broadcast */
int mmvae_mean_over_time_fwd(const float* x, float* y, int L, int N, int d, mmvae_stream_t stream);
int mmvae_mean_over_time_bwd(const float* dy, float* dx, int L, int N, int d, mmvae_stream_t stream);

/* rows of (L,N,d) summed over time into (N,d): used for the broadcast cross-attention gradient */
int mmvae_sum_over_time(const float* x, float* y, int L, int N, int d, mmvae_stream_t stream);

/* y[r, :] = x[r, :] * m[r]  with rows r = (t,b) of a (T,B,V) tensor written as (B,T,V): the decoder's
 * permute(1,0,2) * mask (decoders.py:722).  in (T*B, V) -> out (B,T,V);  bwd is the inverse scatter. */
int mmvae_permute_mask_fwd(const float* x, const uint8_t* mask, float* y, int T, int B, int V,
                           mmvae_stream_t stream);
int mmvae_permute_mask_bwd(const float* dy, const uint8_t* mask, float* dx, int T, int B, int V,
                           mmvae_stream_t stream);
/* ... the first Tk <= T steps only: out (B,Tk,V); mask stays (B,T).  bwd: dy (B,Tk,V) -> dx (T,B,V), exact zeros from step
 * Tk on.  The reference slices the permuted, masked decoder output to the target's mask length (BaseObjective.recon_loss_fn,
 * models/objectives.py:30-52, behind models/decoders.py:720-722): the two as one launch per direction. */
int mmvae_permute_mask_head_fwd(const float* x, const uint8_t* mask, float* y, int T, int B, int V, int Tk,
                                mmvae_stream_t stream);
int mmvae_permute_mask_head_bwd(const float* dy, const uint8_t* mask, float* dx, int T, int B, int V, int Tk,
                                mmvae_stream_t stream);

/* Standard-normal noise for the reparameterised samples (the reference draws it with torch.distributions rsample:
 * models/mmvae_models.py:363-369).  state = {seed, call counter, ticket} (three uint32 on the device, ticket 0); the
 * launch advances the call counter itself, so a captured graph replays with fresh noise and no host work. */
int mmvae_randn(float* out, long n, uint32_t* state, mmvae_stream_t stream);

/* debug: store the device wall clock (100 MHz ticks) into *slot, in stream order (phase timelines of a graph replay) */
int mmvae_debug_timestamp(long long* slot, mmvae_stream_t stream);
/* debug: one thread spins for `ticks` device wall-clock ticks, then stores {start, end} ticks into slot[0..1] */
int mmvae_debug_spin(long long* slot, long long ticks, mmvae_stream_t stream);

/* ---- Generic convolutions (csrc/conv_generic.hip): any channel counts, K <= 4, stride S, padding P -------------
 * First correct path for Enc_SVHN / Dec_SVHN (models/encoders.py:434-478, models/decoders.py:101-147); plain fp32 FMA.
 * Same conventions as the k4-s2-p1 entry points above: layers emit pre-activations, `in_act` is applied to the
 * input, `ep_mode` to the output (`aux` = the tensor a MUL_* epilogue differentiates through).
 *   Conv2d          w [Cout][Cin][K][K], Hout = (Hin + 2P - K) / S + 1
 *   ConvTranspose2d w [Cin][Cout][K][K], Hout = (Hin - 1) S - 2P + K                                            */
int mmvae_conv2d_generic_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                             int Cin, int Cout, int Hin, int Win, int K, int S, int P, int in_act, int ep_mode,
                             mmvae_stream_t stream);
int mmvae_conv2d_generic_dgrad(const float* dy, const float* w, const float* aux, float* dx, int B, int Cin, int Cout,
                               int Hin, int Win, int K, int S, int P, int ep_mode, mmvae_stream_t stream);
int mmvae_conv2d_generic_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int Cin, int Cout, int Hin,
                               int Win, int K, int S, int P, int x_act, int accumulate, mmvae_stream_t stream);
int mmvae_convT2d_generic_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                              int Cin, int Cout, int Hin, int Win, int K, int S, int P, int in_act, int ep_mode,
                              mmvae_stream_t stream);
int mmvae_convT2d_generic_dgrad(const float* dy, const float* w, const float* aux, float* dx, int B, int Cin, int Cout,
                                int Hin, int Win, int K, int S, int P, int ep_mode, mmvae_stream_t stream);
int mmvae_convT2d_generic_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int Cin, int Cout,
                                int Hin, int Win, int K, int S, int P, int x_act, int accumulate,
                                mmvae_stream_t stream);
/* y = sigmoid(x); dx = dy y (1 - y)   (Dec_MNIST, models/decoders.py:266; backward of MMVAE_EP_SIGMOID outputs) */
int mmvae_sigmoid_fwd(const float* x, float* y, long n, mmvae_stream_t stream);
int mmvae_sigmoid_bwd(const float* dy, const float* y, float* dx, long n, mmvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser + utilities
 * ---------------------------------------------------------------------------------------------- */
/* torch.optim.Adam(amsgrad=True) over one flat buffer (models/trainer.py:79-81); step is 1-based and is
 * read from *step_dev (device int) when step_dev != NULL, so that a captured hipGraph replays with the
 * right bias correction (mmvae_step_inc bumps it on the stream).  With step < 0, step_dev is a zero-initialised,
 * 8-byte aligned 24-byte block {int count, int ticket, double beta1^count, double beta2^count}: the launch is step
 * count + 1 and stores the new count and powers itself when its last workgroup finishes -- no separate launch.  g is multiplied by grad_scale first (1/world_size after a sum all-reduce);
 * zero_grad != 0 clears g. */
int mmvae_adam_amsgrad_flat(float* p, float* g, float* m, float* v, float* vmax, long n, float lr, float beta1,
                            float beta2, float eps, int step, int* step_dev, float grad_scale,
                            int zero_grad, mmvae_stream_t stream);
/* `optimizer: adabelief` (models/trainer.py:82-86: adabelief_pytorch.AdaBelief(lr, eps=1e-16, betas=(0.9, 0.999),
 * weight_decouple=True, rectify=False), weight_decay 0): m = b1 m + (1-b1) g; s = b2 s + (1-b2)(g-m)^2 + eps;
 * p -= lr/bc1 * m / (sqrt(s)/sqrt(bc2) + eps).  The package is not vendored by the reference and absent here: restated
 * from Zhuang et al. 2020 (Algorithm 2) and the package's update order, parity unpinned.  step / step_dev / grad_scale /
 * zero_grad as mmvae_adam_amsgrad_flat. */
int mmvae_adabelief_flat(float* p, float* g, float* m, float* s, long n, float lr, double beta1, double beta2, float eps,
                         int step, int* step_dev, float grad_scale, int zero_grad, mmvae_stream_t stream);
int mmvae_step_inc(int* step_dev, mmvae_stream_t stream);
/* dst[i] (+)= sum_r src[r*stride + i],  i < len */
int mmvae_reduce_rows(const float* src, float* dst, int n_rows, long len, long stride, int accumulate,
                      mmvae_stream_t stream);
int mmvae_fill(float* p, long n, float value, mmvae_stream_t stream);

/* Deferred gradient reduction.  With accumulate == MMVAE_ACC_DEFER the *_wgrad / *_bwd_weight / layernorm /
 * embed backward entry points only leave their split partials in `ws` (which must then be private to that call
 * until the reduction runs); the *_layout queries describe them, and ONE mmvae_reduce_segments launch at the end
 * of backward adds every registered segment into its destination:  dst[i] += sum_r src[r*stride + i]. */
#define MMVAE_ACC_DEFER 2
#define MMVAE_MAX_SEGMENTS 64
typedef struct {
  const float* src[MMVAE_MAX_SEGMENTS];
  float* dst[MMVAE_MAX_SEGMENTS];
  int rows[MMVAE_MAX_SEGMENTS];
  int len[MMVAE_MAX_SEGMENTS];
  int stride[MMVAE_MAX_SEGMENTS];
  int blk0[MMVAE_MAX_SEGMENTS]; /* filled by the library */
  int next[MMVAE_MAX_SEGMENTS]; /* filled by the library: chain of segments sharing one destination */
  int n;
} mmvae_reduce_segments_t;
int mmvae_reduce_segments(const mmvae_reduce_segments_t* table, mmvae_stream_t stream);
/* ... with the ELBO assembly of mmvae_lincomb_rowptrs_fwd (out[k] = sum_n W[k][n] sum_b rows[n][b], no d_unit) riding
 * along as one extra workgroup of the same launch: the logged loss values cost no launch in the step's serial tail */
int mmvae_reduce_segments_lincomb(const mmvae_reduce_segments_t* table, const mmvae_rowptrs_t* rows,
                                  const float* W_host, float* out, int n_rows, int B, int n_out,
                                  mmvae_stream_t stream);
/* The fold and the optimiser step of a one-GPU training step as ONE launch: the gradient of element i is
 * g[i] + (sum of the registered partials that target it); every destination must lie inside [g, g + n), destination
 * ranges must not overlap (equal ranges are chained).  Same arithmetic, in the same order, as mmvae_reduce_segments
 * followed by mmvae_adam_amsgrad_flat(step = -1): bit-identical parameters and optimiser state.  step_dev as above
 * (the kernel advances it); rows == NULL: no ELBO assembly rider.  Replaces torch.optim.Adam.step()
 * (reference models/trainer.py:79-81) behind the backward pass's split weight gradients. */
int mmvae_adam_fold_flat(float* p, float* g, float* m, float* v, float* vmax, long n, float lr, float beta1,
                         float beta2, float eps, int* step_dev, float grad_scale, int zero_grad,
                         const mmvae_reduce_segments_t* table, const mmvae_rowptrs_t* rows, const float* W_host,
                         float* out, int n_rows, int B, int n_out, mmvae_stream_t stream);
/* One step's fold + update in SEVERAL launches: elements [lo, hi) of the flat buffers only (lo % 4 == 0), with those
 * segments of `table` whose destination lies inside the range (segments outside are skipped, one that straddles lo or hi
 * is MMVAE_ERR_ARG; table may be NULL).  Every launch of a step reads the same step number from step_dev; exactly one --
 * the LAST in stream order -- passes advance_step = 1 and closes the step.  Per element bit-identical to
 * mmvae_adam_fold_flat over the whole buffer.  (The captured training step updates the decoders' + prior's range beside the
 * encoders' backward pass, so that only the encoders' parameters are left for the serial launch at the end of the step.)
 * Same reference interface as mmvae_adam_fold_flat: torch.optim.Adam.step(), models/trainer.py:79-81. */
int mmvae_adam_fold_range(float* p, float* g, float* m, float* v, float* vmax, long n, long lo, long hi,
                          int advance_step, float lr, float beta1, float beta2, float eps, int* step_dev,
                          float grad_scale, int zero_grad, const mmvae_reduce_segments_t* table,
                          const mmvae_rowptrs_t* rows, const float* W_host, float* out, int n_rows, int B, int n_out,
                          mmvae_stream_t stream);
/* partial layouts: rows x rowlen floats in ws; weight part at column 0, bias part at column bias_col */
int mmvae_conv_wgrad_layout(int B, int Csmall, int Clarge, int Hsmall, int* rows, int* rowlen, int* bias_col);
/* linear weight gradient: `splits` partial (N*K) slabs followed by `splits` partial (N) bias rows; splits == 1
 * means the kernel accumulated directly (nothing to reduce) */
int mmvae_linear_bwd_weight_splits(int M, int N, int K);
/* ... and by mmvae_linear_bwd (1 = written straight to dW/db: <= 256 rows with N % 4 == 0 take the register-operand
 * kernel, which needs dy 16-byte aligned) */
int mmvae_linear_bwd_splits(int M, int N, int K);
int mmvae_layernorm_bwd_rows(int rows, int d); /* partial rows of length 2*d: [dgamma | dbeta] */
int mmvae_embed_bwd_rows(int B, int T, int V);  /* partial rows of length 4 */

#ifdef __cplusplus
}
#endif
#endif /* MMVAE_HIP_H */
