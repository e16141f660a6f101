/*
 * wae.h -- C ABI of libwae_hip.so: the MI355X (gfx950) kernels behind the WaveNet-autoencoder hot path.
 *
 * The reference (MingjieChen/wavenet_autoencoders) has no FFI: its hot path is plain PyTorch modules.
 * Each entry point below replaces the ATen op sequence of the cited reference lines; the Python host
 * (wavenet_autoencoders_amd/) mirrors the reference's module API and binds these with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless the name ends in _host
 *   - nothing here allocates, frees, synchronises or owns memory; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream)
 *   - returns WAE_OK (0) or a negative error code; wae_last_error() gives a thread-local message
 *   - dtype: WAE_F32 (fp32 storage, exact-fp32 MFMA 32x32x2), WAE_BF16 (bf16 storage, MFMA 32x32x16, fp32 accumulate) or
 *     WAE_F16 (fp16 storage, MFMA 32x32x16 f16, fp32 accumulate; same packing as bf16; backward on loss-scaled gradients:
 *     the caller folds the scale into inv_count / ext_dy and 1/scale into the alpha of the weight-gradient tables).
 *     Skip accumulators, biases, losses and the encoder/VQ are always fp32.
 *   - activation layout inside the decoder stack is time-major rows, channels innermost:
 *     x[b][t][Cp] with Cp = channels padded to a multiple of 128 (64 for c; pad channels are zero).  The
 *     reference's (B,C,T) tensors enter/leave through wae_to_btc / wae_from_btc, the first-conv gather
 *     and the head (which writes (B,O,T) logits).
 *   - packed weights are in MFMA A-fragment order produced by wae_pack_gather from index maps the host
 *     builds once (wavenet_autoencoders_amd/packing.py documents the order).
 */
#ifndef WAE_H
#define WAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAE_OK 0
#define WAE_EINVAL (-1)
#define WAE_EUNSUPPORTED (-2)
#define WAE_EHIP (-3)

#define WAE_F32 0
#define WAE_BF16 1
#define WAE_F16 2   /* fp16 storage, MFMA 32x32x16 f16, fp32 accumulate (BASELINE config C5); same packing as WAE_BF16 */

/* flags of wae_glu_desc.flags */
#define WAE_GLU_SAVE_Z 2    /* also store the pre-activation z (B,T,2Hp) for backward */
#define WAE_GLU_NO_OUT 4    /* do not compute/store x' (last layer: the reference's x' is dead, wavenet.py:205-207) */
#define WAE_GLU_CG2 16      /* 16-bit dtypes: 4 waves x 64 time columns (one wave per SIMD), every weight fragment read from LDS feeds
                               two MFMAs -- half the LDS read traffic of the default shape; same results bit for bit */
#define WAE_GLU_PAIR 32     /* GEMM 1 meets at a workgroup barrier on every SECOND weight chunk (four ring slots, weights two chunks ahead): the
                               waves of a SIMD drift by up to one chunk, so one wave's chunk bookkeeping runs under the other's MFMAs;
                               same results bit for bit */
#define WAE_GLU_WAVES4 8    /* bf16: 128-step tiles, 4 waves, two workgroups per CU instead of one 256-step / 8-wave workgroup
                               (same results bit for bit; an A/B switch per launch, not process state) */
#define WAE_GLU_GENERIC 64   /* 16-bit dtypes: run the run-time-scheduled kernel (csrc/glu_fwd.hip) even where a static-schedule
                               instantiation (csrc/glu_fwd_static.hip) exists for the geometry; same results bit for bit */

const char* wae_version(void);
const char* wae_last_error(void);

/* ---- K14 weight norm (modules.py:18, upsample.py:44): w = g * v / ||v|| per output row --------------
 * Parameters live in one flat fp32 arena.  eff <- copy of params, then for every weight-normed row r:
 * eff[v_off[r] .. +cols[r]) = params[g_off[r]] * v / ||v||.  Tables are int64 device arrays of nrows, rows SORTED by v_off and
 * not overlapping (the kernels copy what lies between consecutive rows through instead of copying the whole arena first). */
int wae_weight_norm_fwd(const float* params, float* eff, int64_t n_params, const int64_t* v_off,
                        const int64_t* g_off, const int32_t* cols, int32_t nrows, void* stream);
/* backward: given d_eff (grad wrt effective weights, same layout) accumulate into grads:
 * grads[v] = g/||v|| * (dw - v * (dw.v)/||v||^2), grads[g] = (dw.v)/||v||; every other slot: grads = d_eff. */
int wae_weight_norm_bwd(const float* params, const float* d_eff, float* grads, int64_t n_params,
                        const int64_t* v_off, const int64_t* g_off, const int32_t* cols, int32_t nrows,
                        void* stream);
/* the same over the arena slice [lo, hi) and the weight-normed rows [row_lo, row_hi) that lie inside it: data-parallel
 * training hands the decoder layers' gradients to the all-reduce while the front end's backward is still running
 * (replaces the one-shot gather / reduce of vqwae_train.py:698-706) */
int wae_weight_norm_bwd_range(const float* params, const float* d_eff, float* grads, int64_t lo, int64_t hi,
                              const int64_t* v_off, const int64_t* g_off, const int32_t* cols, int32_t row_lo,
                              int32_t row_hi, void* stream);

/* dst[i + b*dst_stride] = (dtype) (map[i] < 0 ? 0 : src[map[i] + b*src_stride]),  i < n, b < nbatch */
int wae_pack_gather(const float* src, const int32_t* map, void* dst, int64_t n, int32_t nbatch,
                    int64_t src_stride, int64_t dst_stride, int32_t dtype, void* stream);
/* inverse (gradients): dst[map[i] + b*dst_stride] += src[s(i) + b*src_stride] (fp32 atomics; several i may share a
 * slot unless unique == 1, which promises one source per slot and uses plain adds); s(i) = i, or with src_cols > 0 the
 * element (i / src_cols, i % src_cols) of a row-major tile of pitch src_ld.  unique == 2: every row of the tile feeds the
 * one slot map[row * src_cols] (bias gradients from the per-clip ones columns): the row is summed and added once. */
int wae_unpack_scatter_add(const float* src, const int32_t* map, float* dst, int64_t n, int32_t nbatch,
                           int64_t src_stride, int64_t dst_stride, int32_t src_cols, int64_t src_ld, int32_t unique,
                           void* stream);

/* Several of either in ONE launch (a train step packs / scatters eleven families; the reference holds weights in one layout and
 * needs neither).  jobs_host: HOST array of at most WAE_MULTI_MAX jobs, copied into the kernel arguments; the fields mean what
 * the arguments of the single-job entries mean.  The scatter jobs of one call must write disjoint slots when unique != 0.
 * A gather job with map == NULL is a FILL: n fp32 zeros at dst (16-byte aligned; src / strides / nbatch unused) -- the backward's two
 * gradient accumulators are cleared by the launch that packs its weights (round 6). */
#define WAE_MULTI_MAX 16
typedef struct wae_gather_job {
  const float* src;
  const int32_t* map;
  void* dst;
  int64_t n, src_stride, dst_stride;
  int32_t nbatch, dtype;
} wae_gather_job;
typedef struct wae_scatter_job {
  const float* src;
  const int32_t* map;
  float* dst;
  int64_t n, src_stride, dst_stride, src_ld;
  int32_t nbatch, src_cols, unique, pad_;
} wae_scatter_job;
int wae_pack_gather_multi(const wae_gather_job* jobs_host, int32_t njobs, void* stream);
int wae_unpack_scatter_add_multi(const wae_scatter_job* jobs_host, int32_t njobs, void* stream);

/* ---- a1 encoder block (vqvae_model.py:17-23): y = relu(conv1d(x,w,b,stride,pad=k/2)) (+x) ; fp32 (B,C,T)
 * relu/residual/pad selectable so the same entry serves Encoder.lin (vqvae_model.py:50, k=1) and the
 * upsample net's conv_in (upsample.py:77-78: k = 2*cin_pad+1, pad 0, no bias).  Tout = (Tin+2*pad-k)/stride+1 */
int wae_enc_conv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin,
                     int32_t Tin, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t relu,
                     int32_t residual, void* stream);

/* ---- a2 VectorQuantize.forward (vector_quantization.py:21-49) -----------------------------------------
 * lat (B,D,Tq) fp32, emb (K,D).  idx int64 (B*Tq) first-minimum of ||e||^2+||x||^2-2x.e; quant (B,D,Tq);
 * stats[0] = vq_loss = (beta+1)*mean((q-x)^2) forward value, stats[1] = perplexity; hist: (K+1) int32 scratch. */
int wae_vq_nearest(const float* lat, const float* emb, int64_t* idx, float* quant, float* stats, int32_t* hist,
                   int32_t B, int32_t D, int32_t Tq, int32_t K, float beta, void* stream);

/* ---- section 8(f) rank 3: the sliced / EMA quantizers (vector_quantization.py:51-128, :132-235, :239-306) -----------
 * wae_vq_slice: the search above on channels [d0, d0+D) of a (B,Dtot,Tq) tensor against that slice's own codebook
 * emb (K,D); quant is the (B,Dtot,Tq) output (only the slice is written).  mode 0: search + gather (:84-110),
 * 1: search only (idx, hist, stats[1]), 2: gather with the idx given (the EMA classes gather from the codebook AFTER
 * its update, :217-218,:292).  stats[0] = c_loss * mean((q-x)^2) over the slice, stats[1] = slice perplexity
 * (the sliced classes ADD the slice perplexities, :125-127).  hist: (K+1) int32 scratch; its K counts feed
 * wae_vq_ema_update.
 * wae_vq_ema_update (training branch, :190-215 / :275-290): cluster_size = decay*cluster_size + (1-decay)*counts,
 * Laplace smoothing (+1e-5, K*1e-5), ema_w = decay*ema_w + (1-decay)*sum of the latents assigned to each code,
 * emb = ema_w / cluster_size.  All three buffers are updated in place.
 * wae_vq_slice_bwd: dlat = dquant + c_lat*(x-q); demb[idx] += c_emb*(q-x) (demb NULL for EMA codebooks). */
int wae_vq_slice(const float* lat, const float* emb, int64_t* idx, float* quant, float* stats, int32_t* hist, int32_t B,
                 int32_t Dtot, int32_t d0, int32_t D, int32_t Tq, int32_t K, float c_loss, int32_t mode, void* stream);
int wae_vq_ema_update(const float* lat, const int64_t* idx, const int32_t* hist, float* ema_cluster_size, float* ema_w,
                      float* emb, int32_t B, int32_t Dtot, int32_t d0, int32_t D, int32_t Tq, int32_t K, float decay,
                      void* stream);
int wae_vq_slice_bwd(const float* lat, const float* quant, const int64_t* idx, const float* dquant, float* dlat,
                     float* demb, int32_t B, int32_t Dtot, int32_t d0, int32_t D, int32_t Tq, float c_lat, float c_emb,
                     void* stream);

/* ---- a3 one upsample stage (upsample.py:19-21 stretch + :39-46 FIR) ------------------------------------
 * in (B,C,Tin) fp32 -> nearest-stretch by s, FIR w[2s+1] zero-padded.  If out_btc != 0 the result is written
 * time-major (B,Tin*s,Cp) in `dtype` (pad channels zeroed), else (B,C,Tin*s) fp32. */
int wae_upsample_stage_fwd(const float* in, const float* w, void* out, int32_t B, int32_t C, int32_t Tin,
                           int32_t s, int32_t out_btc, int32_t Cp, int32_t dtype, void* stream);

/* ---- a4/K5 hoisted global conditioning: zb[b][l][2Hp] = bias_l + Wg_l . g_b  (modules.py:148-152) ------
 * g_b = eff[emb_off + gid[b]*Cg ..] (Embedding lookup, wavenet.py:185-190) when gid != NULL, else gvec[b*Cg ..]
 * (external features); both NULL or wg_off < 0 = no global conditioning.  Layer l reads its conv bias at
 * eff[bias_off + l*layer_stride] and its conv1x1g weight (G,Cg) at eff[wg_off + l*layer_stride]. */
int wae_gproj_fwd(const float* eff, int64_t wg_off, int64_t bias_off, int64_t layer_stride, const int32_t* gid,
                  int64_t emb_off, const float* gvec, float* zb, int32_t B, int32_t L, int32_t G, int32_t Hp,
                  int32_t Cg, int32_t n_speakers, int32_t* err, void* stream);

/* backward of wae_gproj_fwd.  c1: the per-layer dW1 tiles of wae_gemm_tn_tiles (L x 2Hp x ld fp32) whose columns
 * ones_col + b hold sum_t dz[b,t,row] = d loss / d zb[b][l][row]; adds into d_eff: conv bias, conv1x1g weight and
 * (gid != NULL) the embedding rows. */
int wae_gproj_bwd(const float* eff, float* d_eff, int64_t wg_off, int64_t bias_off, int64_t layer_stride,
                  const int32_t* gid, int64_t emb_off, const float* gvec, const float* c1, int64_t c_layer_stride,
                  int64_t ld, int32_t ones_col, int32_t B, int32_t L, int32_t G, int32_t Hp, int32_t Cg,
                  int32_t n_speakers, void* stream);

/* ---- a5 first_conv on one-hot input = column gather + bias (wavenet.py:119-122,203) --------------------
 * idx (B*T) int32 class ids; table (O,Rp) fp32 = W^T ; x0 (B,T,Rp) dtype.  scalar mode: xs (B*T) fp32,
 * table (1,Rp). */
int wae_first_conv_fwd(const int32_t* idx, const float* xs, const float* table, const float* bias, void* x0,
                       int64_t BT, int32_t Rp, int32_t O, int32_t dtype, int32_t* err, void* stream);

/* One-hot rows of the class ids: out (n, width) dtype, out[i][ids[i]] = 1 (ids outside [0, width): a zero row).  The operand that
 * turns the first conv's weight gradient (autograd of wavenet.py:203: dW[:, class] = sum of dx0 rows whose input is `class`) into a
 * P^T Q contraction of wae_gemm_tn_static. */
int wae_onehot_rows(const int32_t* ids, void* out, int64_t n, int32_t width, int32_t dtype, void* stream);

/* Ids that index tables (class ids -> first-conv table / targets, speaker ids -> embedding rows): the reference raises
 * IndexError (nn.Embedding, wavenet.py:185-190; one-hot encoding, vqwae_train.py:511).  The kernels clamp an id outside its
 * table -- no access leaves it -- and OR a WAE_ERR_* bit into the caller's sticky device word `err` (nullable); wae_check_ids
 * does the same for any id array (e.g. the CE targets) against [lo, hi).  The host raises when it next reads the word. */
#define WAE_ERR_CLASS_ID 1
#define WAE_ERR_SPEAKER_ID 2
#define WAE_ERR_TARGET_ID 4
int wae_check_ids(const int32_t* ids, int64_t n, int32_t lo, int32_t hi, int32_t* err, int32_t code, void* stream);
/* One-hot (B, C, T) fp32 input -> class ids (B, T).  wavenet.py:203 applies first_conv to any (B, C, T) float tensor; the kernels
 * gather weight rows by class id, the same arithmetic for one-hot columns only.  A column that is not exactly one 1.0 among zeros
 * ORs `code` (WAE_ERR_NOT_ONEHOT) into `err`: the host refuses dense inputs instead of arg-maxing them.  Strides in elements. */
#define WAE_ERR_NOT_ONEHOT 8
int wae_onehot_to_ids(const float* x, int32_t B, int32_t C, int32_t T, int64_t stride_b, int64_t stride_c, int64_t stride_t,
                      int32_t* ids, int32_t* err, int32_t code, void* stream);

/* ---- two chains of layer launches (round 6) ------------------------------------------------------------------
 * Every workgroup of a layer launch stores its outputs at the same time (forward: z, u, x' -- 106 MB at BASELINE C2, ~20 us of a 55-us
 * launch in which nothing computes).  One launch cannot be de-phased against itself, two CHAINS of launches can: the host runs the
 * gated stack (and the backward sweep) as two half-batch chains on two streams -- the same entry points with B / 2 clips and offset
 * pointers; every array of this ABI is clip-major -- and starts the second chain `us` microseconds late with this call: one wave that
 * sleeps on the wall clock.  Results are bit for bit those of one chain (no clip reads another clip's rows). */
int wae_stream_delay(double us, void* stream);

/* ---- a6 ResidualConv1dGLU._forward (modules.py:115-163) -------------------------------------------------
 * One fused layer: dilated causal conv + 1x1(c) + hoisted 1x1(g) + gate + 1x1 out + residual.  The skip 1x1
 * (modules.py:157) and `skips += h` (wavenet.py:204-207) are deferred: the layer stores its gated activation
 * u_l = tanh(a)*sigmoid(b) and wae_head_fwd contracts all layers' u against [W_skip_0 .. W_skip_{L-1}] at once. */
typedef struct wae_glu_desc {
  int32_t dtype;
  int32_t B, T;
  int32_t Rp, Ccp, Hp; /* padded: Rp % 128 == 0, Ccp % 64 == 0 (may be 0), Hp % 32 == 0, Hp <= 192 */
  int32_t ktaps;       /* kernel_size */
  int32_t dilation;
  int32_t flags;
} wae_glu_desc;
/* x_in,x_out (B,T,Rp) dtype; c_up (B,T,Ccp) dtype; u_out: this layer's Hp columns inside a (B,T,u_stride) dtype
 * buffer (pointer already offset to the layer's first column); zb (B,2Hp) fp32 for THIS layer (row stride
 * zb_stride floats) = conv bias + hoisted global conditioning; z_save (B,T,2Hp) dtype or NULL;
 * w_packed = [W1 chunks | W_out chunks] in fragment order; bias_out (Rp) fp32. */
int wae_glu_layer_fwd(const wae_glu_desc* d, const void* x_in, void* x_out, const void* c_up, void* u_out,
                      int64_t u_stride, const float* zb, int64_t zb_stride, void* z_save, const void* w_packed,
                      const float* bias_out, void* stream);
/* The same layer in TRAINING with dropout > 0 (modules.py:127-128: F.dropout in front of the dilated convolution only; the
 * residual path keeps the layer's input): x_conv = wae_dropout_fwd(x_in) is the operand of the convolution taps, x_in that
 * of the residual add.  wae_glu_layer_fwd is this entry with x_conv = x_in. */
int wae_glu_layer_fwd_drop(const wae_glu_desc* d, const void* x_in, const void* x_conv, void* x_out, const void* c_up, void* u_out,
                           int64_t u_stride, const float* zb, int64_t zb_stride, void* z_save, const void* w_packed,
                           const float* bias_out, void* stream);
/* xd[i] = keep(seed, i) ? x[i] / (1 - p) : 0 over n elements of `dtype`; keep = a counter-based hash of (seed, i) whose top 24
 * bits are >= p * 2^24 (restated by the oracle).  Backward of the convolution input through the same mask, fused with the
 * residual add of modules.py:161: out[i] = alpha * (g_next[i] + (keep ? acc[i] / (1 - p) : 0)), acc = the tap contraction
 * W1^T dz (wae_gemm_tm mode 0); the weight gradient dW1 contracts dz against xd. */
int wae_dropout_fwd(const void* x, void* xd, int64_t n, uint64_t seed, float p, int32_t dtype, void* stream);
int wae_dropout_bwd(const void* acc, const void* g_next, void* out, int64_t n, uint64_t seed, float p, float alpha, int32_t dtype,
                    void* stream);
int64_t wae_glu_packed_bytes(const wae_glu_desc* d);

/* ---- a7+a8+a9 skip sum + head (wavenet.py:204-214) + MaskedCrossEntropyLoss (vqwae_train.py:363-379,:764) ---
 * skips = sum_l (W_skip_l u_l + b_skip_l);  h0 = relu(skips * scale);  h1 = relu(W1 h0 + b1);  logits = W3 h1 + b3.
 * u (B,T,Ku) dtype holds all layers' gated activations, Ku = L*Hp rounded up to 64 (pad columns zero).
 * bias = [sum_l b_skip_l (Sp) | b1 (Sp) | b3 (Op)] fp32.  logits (B,O,T) fp32 or NULL.
 * If target != NULL: nll[b*T+t] = logsumexp(logits[:,t]) - logits[target[b*T+t+1], t] for t < T-1
 * (the reference's one-step shift), 0 at t = T-1; lse (B,T) or NULL receives logsumexp(logits[:,t]).
 * h0_save/h1_save (B,T,Sp) dtype or NULL (for backward). */
typedef struct wae_head_desc {
  int32_t dtype;
  int32_t B, T;
  int32_t Ku;     /* columns of u, multiple of 64 */
  int32_t Sp, Op; /* padded to multiples of 128 */
  int32_t O;      /* true class count */
  float scale;    /* sqrt(1/L) */
} wae_head_desc;
int wae_head_fwd(const wae_head_desc* d, const void* u, const void* w_packed, const float* bias, float* logits,
                 const int32_t* target, float* nll, float* lse, void* h0_save, void* h1_save, void* stream);

/* The second half of wae_head_fwd -- GEMM 1, GEMM 2, logits and / or the fused cross-entropy (wavenet.py:208-214,
 * vqwae_train.py:363-379,:764) -- from a stored h0 = relu(sqrt(1/L) (sum_l b_skip_l + W_skip u)) (B,T,Sp) dtype, for callers that ran
 * the skip contraction as a wae_gemm_tm launch (mode 3, the first Ku/CK chunks of the same packed stream as its weights).  w_tail =
 * w_packed + (Ku / CK) * (Sp / 32) * 4096 bytes (CK = 64 for 16-bit, 32 for fp32); bias as in wae_head_fwd (first Sp entries unread). */
int wae_head_fwd_from_h0(const wae_head_desc* d, const void* h0, const void* w_tail, const float* bias, float* logits,
                         const int32_t* target, float* nll, float* lse, void* h1_save, void* stream);
/* backward of the head down to dskip (csrc/head_bwd.hip).  CE mode (ext_dy == NULL): logits are recomputed from h1,
 * dy = (softmax - onehot(target[t+1])) * [t < len-1] * inv_count with the forward's lse (B,T); DMoL mode: ext_dy
 * (B,T,Op) dtype is the loss gradient.  Outputs, all time-major dtype: dy_out (B,T,Op), dh1_out (B,T,Sp) (pre-ReLU
 * gradient of h1), dskip_out (B,T,Sp) = d loss / d skips.  w_packed = [W3 first-order | W3^T | W1^T second-order]. */
int wae_head_bwd(const wae_head_desc* d, const void* h0, const void* h1, const void* w_packed, const float* b3,
                 const float* lse, const int32_t* target, const int32_t* lengths, float inv_count, const void* ext_dy,
                 void* dy_out, void* dh1_out, void* dskip_out, void* stream);
int64_t wae_head_bwd_packed_bytes(const wae_head_desc* d);
int64_t wae_head_packed_bytes(const wae_head_desc* d);

/* out[i] = sum_{l<L} src[off + l*stride + i] for i < n, 0 for n <= i < n_pad  (sum of the skip biases) */
int wae_sum_rows(const float* src, int64_t off, int64_t stride, int32_t L, int32_t n, int32_t n_pad, float* out,
                 void* stream);

/* masked mean of per-sample losses: out[0] = sum_{b,t<len[b]-1} nll / sum mask  (vqwae_train.py:379) */
int wae_masked_mean(const float* nll, const int32_t* lengths, float* out, int32_t B, int32_t T, void* stream);
/* The criterion object of vqwae_train.py:363-379 on EXPLICIT logits (the drop-in MaskedCrossEntropyLoss; training proper uses
 * the CE fused into wae_head_fwd): logits (B,C,T) fp32, target (B,T) int64 -> nll (B,T), lse (B,T); backward
 * dlogits = (softmax - onehot) * w[b,t].  A target outside [0, C) is clamped and flagged in err (WAE_ERR_TARGET_ID).
 * wae_weighted_mean: out[0] = sum(v*m) / sum(m), out[1] = sum(m) for any mask m (:374-379). */
int wae_ce_logits_fwd(const float* logits, const int64_t* target, float* nll, float* lse, int32_t B, int32_t C, int32_t T,
                      int32_t* err, void* stream);
int wae_ce_logits_bwd(const float* logits, const int64_t* target, const float* lse, const float* w, float* dlogits, int32_t B,
                      int32_t C, int32_t T, void* stream);
int wae_weighted_mean(const float* v, const float* m, int64_t n, float* out, void* stream);

/* ---- a10 discretized mixture of logistics (mixture.py:26-106; wrapper vqwae_train.py:382-401, shift :766) --
 * y_hat (B,3M,T) fp32 = [logit pi | mu | log s]; y (B,T) fp32 in [-1,1].  nll[b,t] = -log p(y[b,t+shift] | y_hat[b,:,t])
 * (0 where t+shift >= T); dy_hat (B,3M,T) or NULL = d nll[b,t] / d y_hat[b,:,t]. */
int wae_dmol_loss_fwd(const float* y_hat, const float* y, float* nll, float* dy_hat, int32_t B, int32_t M,
                      int32_t T, int32_t num_classes, float log_scale_min, int32_t shift, void* stream);
/* ---- a11 sample_from_discretized_mix_logistic (mixture.py:118-156) with caller-supplied U(1e-5,1-1e-5) draws:
 * y (B,3M,Tn), u_mix (B,Tn,M), u_log (B,Tn) -> out (B,Tn) in [-1,1]. */
int wae_dmol_sample(const float* y, const float* u_mix, const float* u_log, float* out, int32_t B, int32_t M,
                    int32_t Tn, float log_scale_min, int32_t clamp_log_scale, void* stream);

/* ---- a15 clip_grad_norm_ + Adam + EMA over the flat arena (vqwae_train.py:776-787, :339-350) ---------------
 * coef = min(1, clip/(||g||+1e-6)) (clip <= 0: off); torch.optim.Adam update with bias correction for the
 * 1-based `step`; shadow -= (1-ema_decay)*(shadow-p) when shadow != NULL.  scratch: one double.
 * grad_norm_out (1 float) or NULL receives the pre-clip global L2 norm. */
int wae_clip_adam_ema(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* shadow, int64_t n,
                      double* scratch, float* grad_norm_out, int32_t step, double lr, double beta1, double beta2,
                      double eps, double weight_decay, double clip_thresh, double ema_decay, void* stream);

/* ---- softmax over the channel dimension of (B, C, T) fp32 logits: WaveNet.forward(softmax=True) / VQVAE.forward(softmax=True)
 * (wavenet.py:214, vqvae_model.py:79-80: F.softmax(x, dim=1)) and its backward dx = p (dp - sum_c p dp).  p may alias x; dx may alias dp. */
int wae_softmax_bct_fwd(const float* x, float* p, int32_t B, int32_t C, int32_t T, void* stream);
int wae_softmax_bct_bwd(const float* p, const float* dp, float* dx, int32_t B, int32_t C, int32_t T, void* stream);

/* C[n] = alpha * A[n] (M x K, row stride lda) B[n] (K x N, ldb), fp32, n < nbatch with the given batch strides: the per-layer products
 * sqrt(.5) W1_cur[l] W_out[l-1] a caller of wae_ar_generate_coop_fused forms once per weight update (engine.py: _pack_ar_fused). */
int wae_bmm_f32(const float* a, const float* b, float* c, int32_t nbatch, int32_t M, int32_t K, int32_t N, int64_t lda, int64_t ldb,
                int64_t ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c, float alpha, void* stream);

/* ---- a12 incremental (autoregressive) decoding: Conv1d.incremental_forward (conv.py:17-62) and
 * WaveNet.incremental_forward (wavenet.py:218-346) as ONE persistent launch, one workgroup per utterance ------
 * mode 0: teacher-forced (the reference's test_inputs, softmax=False, quantize=False): logits out, inputs consumed
 * mode 1: greedy, the argmax class is fed back          mode 2: categorical draw by inverse CDF from uniforms[b,t]
 * mode 3 / 4 (wae_ar_generate only; quantize=False, wavenet.py:335-338 skipped): the softmax probabilities / the raw logits
 *   of step t are the dense decoder input of step t+1 (first_conv on a (1, O) row) and the step's row of out_logits
 * Partial teacher forcing (test_inputs shorter than T, wavenet.py:300-305): desc.n_forced.
 * Weights are blocked [k/EPL][rows padded to 64][EPL] (EPL = 8 bf16 / 4 fp32), per layer [W1 (G x (k*R+Cc)) | W2
 * ((R+S) x H)] with layer_stride_bytes between layers; head = [S x S | O x S].  ring: B x ring_total floats of
 * per-layer history, ring_off[l] = float offset of layer l ((k-1)*d_l+1 rows of R).  zb as in wae_gproj_fwd.
 * first_tab (O, Rp) / first_bias as in wae_first_conv_fwd; c_up (B,T,Ccp) already upsampled (wavenet.py:276-280).
 * inputs (B,T) int32 class ids or NULL (then init_idx starts every utterance, wavenet.py:288). */
typedef struct wae_ar_desc {
  int32_t dtype;
  int32_t B, T;
  int32_t L, R, Rp, G, Hp, S, O, Cc, Ccp, ktaps;
  int32_t mode;
  int32_t init_idx;
  int32_t scalar_input; /* 0: wae_ar_generate / wae_ar_generate_coop; 1: wae_ar_generate_scalar */
  float scale;          /* sqrt(1/L) */
  int32_t n_forced;     /* with inputs: steps t < n_forced consume inputs[t], later steps the fed-back output
                           (test_inputs shorter than T, wavenet.py:300-305); <= 0 or >= T: every step is forced */
  /* wae_ar_generate_coop only (the library reads no environment variable; rounds 4-5 had three): */
  int32_t coop_generic;  /* 1: the any-shape cooperative kernel also where the reference's geometry has one with its sizes as constants */
  int32_t resident_lds;  /* layers whose weight packets the fast kernel keeps in LDS: 0 = as many as fit, n > 0 = n, < 0 = none */
  int32_t resident_regs; /* ... and in registers (accumulation registers, then hand-allocated arch VGPRs): 0 = all that fit, n > 0 = n,
                            < 0 = none.  Where the packets wait is not arithmetic: results are bitwise the same for every split. */
} wae_ar_desc;
int wae_ar_generate(const wae_ar_desc* d, const int32_t* dilations, const int64_t* ring_off, float* ring,
                    int64_t ring_total, const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes,
                    const float* bias2, const float* zb, const float* first_tab, const float* first_bias,
                    const void* w_head, const float* head_bias, const void* c_up, int32_t c_dtype,
                    const int32_t* inputs, const float* uniforms, int32_t* out_idx, float* out_logits, void* stream);

/* Scalar-input decoders (first_conv has one input channel; wavenet.py:284-285,325-333): the fed-back quantity is the
 * float drawn by sample_from_discretized_mix_logistic (mixture.py:118-156) from the step's 3M mixture parameters, on
 * caller-supplied uniforms u_mix (B,T,M), u_log (B,T) in (1e-5, 1-1e-5).  inputs_f (B,T) teacher-forces the inputs
 * (step t consumes inputs_f[t]; the start value is 0 without it).  out_samples (B,T) and/or out_params (B,3M,T). */
int wae_ar_generate_scalar(const wae_ar_desc* d, const int32_t* dilations, const int64_t* ring_off, float* ring,
                           int64_t ring_total, const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes,
                           const float* bias2, const float* zb, const float* first_tab, const float* first_bias,
                           const void* w_head, const float* head_bias, const void* c_up, int32_t c_dtype,
                           const float* inputs_f, const float* u_mix, const float* u_log, float log_scale_min,
                           int32_t clamp_log_scale, float* out_samples, float* out_params, void* stream);

/* The same decoding with ONE utterance spread over C cooperating workgroups / CUs (csrc/ar_coop.hip): every layer is
 * split by gate channels; the members all-reduce their shares of x' once per layer and of the skip vector once per
 * sample through `acc` (B x wae_ar_coop_acc_floats(d) floats), `msg` ((B, 2, C, NV) 8-byte granules, NV =
 * wae_ar_coop_msg_values(d, C)) carries the start-up handshake; each member keeps its own copy of the history rings:
 * ring is (B, C, ring_total).  B <= 8, C <= 32, R, S and O <= 256.  The caller zeroes msg, acc and error (>= 64 ints) before
 * the launch; error[0] != 0 afterwards means a wait timed out (the output is then invalid).  No atomics: every share is one
 * stored {sequence number, fp32} granule and the members add them in a fixed order -- results are bitwise reproducible.  The
 * reference's geometry (R = G = S = O = 256, 3 taps, Cc <= 256) on C = 32 runs a kernel with those sizes as constants
 * (wae_ar_desc.coop_generic = 1 keeps the any-shape kernel); both zero-fill / overwrite `ring` themselves (the caller should still hand
 * over a zeroed ring: the fast kernel's members zero-fill their shares only when they sit on one XCD).  That kernel's
 * 32 members share ONE history ring per utterance (member 0's region of the (B, C, ring_total) allocation: every member writes every
 * row, the same bits) and, in 16-bit storage, keep every layer's weight packets on chip for the whole clip -- 6 layers in LDS, 11 in
 * the accumulation registers, 3 in hand-allocated arch VGPRs at the reference's 20 layers; deeper stacks stream the rest from L2.
 * wae_ar_desc.resident_lds / resident_regs override the number of layers kept in LDS / in registers (-1 -1: the streaming form;
 * results are bitwise the same for every split, tests/test_gpu_ar.py). */
int64_t wae_ar_coop_acc_floats(const wae_ar_desc* d);
int wae_ar_coop_msg_values(const wae_ar_desc* d, int32_t C);
int wae_ar_generate_coop(const wae_ar_desc* d, int32_t C, const int32_t* dilations, const int64_t* ring_off, float* ring,
                         int64_t ring_total, const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes,
                         const float* bias2, const float* zb, const float* first_tab, const float* first_bias,
                         const void* w_head, const float* head_bias, const void* c_up, int32_t c_dtype,
                         const int32_t* inputs, const float* uniforms, int32_t* out_idx, float* out_logits,
                         uint64_t* msg, float* acc, int32_t* error, void* stream);
/* The same call with one more operand for the reference's geometry on 32 members: w_fused = (L, G, Hp) row-major in the model's element
 * type, row block l = sqrt(.5) * W1_cur[l] . W_out[l-1] (the current-tap columns of layer l's dilated convolution times the previous
 * layer's conv1x1_out; block 0 unused) -- formed once per weight update by the caller (a plain matrix product).  Because
 * z_l = W1_cur[l] x_l + .. and x_l = sqrt(.5) (W_out[l-1] u_{l-1} + b_{l-1} + x_{l-1}), a member's 4 channels of u_{l-1} give its share of
 * all gate rows of layer l directly: ONE reduce-scatter per layer stays on the critical path instead of the all-reduce's two hand-overs
 * (csrc/ar_coop.hip: FUSED).  w_fused = NULL, or any other geometry: exactly wae_ar_generate_coop. */
int wae_ar_generate_coop_fused(const wae_ar_desc* d, int32_t C, const int32_t* dilations, const int64_t* ring_off, float* ring,
                         int64_t ring_total, const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes,
                         const float* bias2, const float* zb, const float* first_tab, const float* first_bias,
                         const void* w_head, const float* head_bias, const void* c_up, int32_t c_dtype,
                         const int32_t* inputs, const float* uniforms, int32_t* out_idx, float* out_logits,
                         uint64_t* msg, float* acc, int32_t* error, const void* w_fused, void* stream);

/* ---- backward data path of the gated stack: C[t][M] = sum_s W_s . X_s[t + shift_s] on time-major operands ----
 * (autograd of modules.py:115-163; see csrc/gemm_tm.hip).  mode 0: out (t, M) = acc.  mode 1 (residual):
 * out = alpha * (acc + aux[t]).  mode 2 (gate backward): acc = du over M = Hp rows, aux = z (t, 2Hp),
 * out (t, 2Hp) = [du*sigmoid(b)*(1-tanh(a)^2) | du*tanh(a)*sigmoid(b)*(1-sigmoid(b))].
 * The four *_host arguments are HOST arrays of nsrc entries (device pointers, strides, cols, shifts).
 * Sources: nsrc <= 4 time-major arrays (row stride in elements, cols a multiple of 64 bf16 / 32 fp32), row t+shift,
 * zero outside the clip.  w_packed: [sum cols / CK] chunks of (M/32) x 4 fragment blocks (first-GEMM order). */
typedef struct wae_tm_desc {
  int32_t dtype;
  int32_t B, T;
  int32_t M;    /* output rows, multiple of 32: M/32 in {1,2,3,4,6,8}, or (modes 0/1/3/4) any multiple of 128: the output is then
                   cut into equal slices of 8, 6 or 4 tiles and w_packed is [slice][chunk] (packing.py: first_gemm_map) */
  int32_t nsrc;
  int32_t mode;
  float alpha;
  int32_t flags; /* WAE_TM_INTERLEAVE: the chunk stream visits the (equally wide) sources round-robin per 128-byte column
                    block -- [block 0 of source 0, of source 1, ..., block 1 of source 0, ...] -- instead of source by source:
                    the dilated taps of one column block are then read back to back, while the tiles d and 2d rows away
                    fetch the same rows into the same L2 (w_packed follows the same order) */
} wae_tm_desc;
#define WAE_TM_INTERLEAVE 1
#define WAE_TM_ONE_WG 2   /* bf16 gate-backward / residual / ReLU-backward launches: one 4-wave workgroup per CU with the 8 KiB
                             staging tiles instead of two per CU (same results; an A/B switch per launch).  It also keeps a mode-1 / -2 / -3
                             launch on this generic kernel where csrc/gemm_tm8.hip (below) has an instantiation. */
/* Mode 3 with ONE source, no shift, 16-bit storage, M a multiple of 256 and an even chunk count (the head's skip contraction, the
 * wide head's h0 / h1 launches) runs on csrc/gemm_tm8.hip since round 6: 8 consumer + 4 loader waves per workgroup, 256 time columns,
 * both operands by LDS-DMA -- same packed stream, same accumulation order, bit-identical outputs (170 against 196 us at C2). */
/* Modes 1 (interleaved taps of ONE array, M a multiple of 256) and 2 (M = Hp in {256, 192, 128}) in 16-bit storage with an even chunk
 * count run on gemm_tm8x_kernel of the same file since round 6: 256 columns per workgroup, both operand streams by LDS-DMA, the
 * eight waves sharing the requests (csrc/glu_bwd8.hip's schedule); bit for bit the generic kernel's results. */
/* (flag value 4 was WAE_TM_BLDS, the 8-wave LDS-staged-operand shape of rounds 3-4: measured not faster, removed in round 5) */
int wae_gemm_tm(const wae_tm_desc* d, const void* const* src_host, const int64_t* src_stride_host,
                const int32_t* src_cols_host, const int32_t* src_shift_host, const void* w_packed, void* out,
                int64_t out_stride, const void* aux, int64_t aux_stride, void* stream);

/* Wide decoder head: skip / head widths above 256 (BASELINE config C5: R = S = 512) do not fit the register-chained
 * wae_head_fwd / wae_head_bwd; the same arithmetic (wavenet.py:204-214, vqwae_train.py:363-379 with the shift of :764)
 * then runs as launches of the kernel above with epilogues, h0 / h1 / dy / dh1 passing through HBM:
 *   mode 3: out = relu(alpha * (bias[m] + acc)), aux = fp32 bias (M floats, aux_stride ignored)
 *   mode 4: out = aux[t][m] > 0 ? alpha * acc : 0, aux = the saved activation (t, M) in the compute dtype
 *   mode 5 (wae_gemm_tm_ce): acc = bias + W3 h1 over M = Op in {128, 256} padded classes; ce->logits (B,O,T) fp32 and / or
 *           ce->nll[b,t] = lse - y[target[t+1]] (0 at t = T-1), ce->lse optional; `out` unused
 *   mode 6 (wae_gemm_tm_ce): out (t, Op) = (exp(bias + acc - ce->lse[t]) - onehot(target[t+1])) * w[t],
 *           w[t] = ce->inv_count if t + 1 < min(lengths[b], T) else 0 (lengths == NULL: T) */
typedef struct wae_tm_ce {
  float* logits;
  const int32_t* target;
  float* nll;
  float* lse;
  const int32_t* lengths;
  float inv_count;
  int32_t O;
} wae_tm_ce;
int wae_gemm_tm_ce(const wae_tm_desc* d, const void* const* src_host, const int64_t* src_stride_host,
                   const int32_t* src_cols_host, const int32_t* src_shift_host, const void* w_packed, void* out,
                   int64_t out_stride, const float* bias, const wae_tm_ce* ce, void* stream);

/* ---- weight gradients: C[m][n] += alpha * sum_{b,t} P[b,t][m] * Q[b,t+shift][n]  (csrc/gemm_tn.hip) ----------
 * The work is described as an array of 128x128 output tiles in DEVICE memory; one launch processes them all (the
 * several weight gradients of a layer share a launch).  P, Q: time-major dtype arrays, pointers already offset to the
 * tile's first column, of which the first m_valid / n_valid columns take part.  onehot != NULL: P[t][m] =
 * (onehot[b*T+t] == m0 + m) (first-conv gradient).  ones_col >= 0: a virtual all-ones Q column at that tile-local
 * index; clip b accumulates sum_t P[b,t][m] into C[m][ones_col + b] (bias / per-clip conditioning-bias gradients).
 * Each tile is contracted over the time range of one clip split `splits` ways; results are added with fp32 atomics
 * (not bitwise reproducible run to run). */
typedef struct wae_tn_tile {
  const void* P;
  const void* Q;
  const int32_t* onehot;
  float* C;
  int64_t p_stride, q_stride, ldc;
  int32_t m_valid, n_valid;
  int32_t m0;
  int32_t shift;
  int32_t ones_col;
  float alpha;
} wae_tn_tile;
int wae_gemm_tn_tiles(int32_t dtype, const wae_tn_tile* tiles_dev, int32_t ntiles, int32_t B, int32_t T, int32_t splits,
                      void* stream);

/* ---- backward data path across one layer boundary, fused (csrc/glu_bwd.hip; autograd of modules.py:115-163) ----------
 *   dx_l-hat = alpha * (dx_{l+1}-hat + sum_tap W1_l,tap^T dz_l[t + (k-1-tap) d])   -> g_out (B,T,Rp), also kept in registers
 *   dz_{l-1} = gate'(z_{l-1}) * (W_out_{l-1}^T dx_l-hat + W_skip_{l-1}^T dskip)    -> dz_prev (B,T,dz_stride)
 * = wae_gemm_tm mode 1 for layer l followed by mode 2 for layer l-1, without re-reading dx_l-hat.  dz / dz_prev point at
 * the layers' 2Hp columns inside the (B,T,dz_stride) buffer; z_prev is (B,T,2Hp); w_x = the mode-1 weights of layer l,
 * w_uo = W_out_{l-1}^T in accumulator-row k order (packing.py: bwd_uo_map), w_us = the W_skip part of the mode-2 weights.
 * wae_glu_bwd_fused_supported(Rp, Hp) tells whether an fp32 instance exists (else use the two wae_gemm_tm launches),
 * wae_glu_bwd_fused_supported16 the same for 16-bit storage.  16-bit storage runs the two-workgroups-per-CU form of round 5
 * (csrc/glu_bwd.hip: glu_bwd_pair_kernel), which takes w_x in the INTERLEAVED chunk order of wae_gemm_tm mode 1 with
 * WAE_TM_INTERLEAVE (packing.py: bwd_x_map) -- dx_l-hat is then bitwise what that launch stores; fp32 takes w_x tap by tap. */
typedef struct wae_glu_bwd_desc {
  int32_t dtype, B, T, Rp, Hp, Sp, ktaps, dilation;
  float alpha;
} wae_glu_bwd_desc;
int wae_glu_bwd_fused_supported(int32_t Rp, int32_t Hp);
int wae_glu_bwd_fused_supported16(int32_t Rp, int32_t Hp);
int wae_glu_bwd_fused(const wae_glu_bwd_desc* d, const void* dz, int64_t dz_stride, const void* g_next, void* g_out,
                      const void* dskip, const void* z_prev, void* dz_prev, const void* w_x, const void* w_uo,
                      const void* w_us, void* stream);
/* 16-bit storage, three taps, Ccp = 64: the same launch with the conditioning gradient folded in (round 5).  dc = sum_l Wc_l^T dz_l
 * (the ONE K = L * 2Hp launch of wae_gemm_tm mode 0 that re-reads every layer's dz) rides on phase A's shift-0 tap, whose operand
 * fragments are dz_l[t]: w_c = layer l's 2Hp / 64 chunks (8 KiB each) of that launch's weight stream (packing.py: bwd_c_map), dc_acc =
 * the fp32 (B,T,64) running sum over the layers, dc_mode bit 0: add dc_acc's previous content (clear: the first launch of a sweep),
 * bit 1: write the sum in the storage dtype to dc_out (B,T,64) instead (the last launch of the sweep).  last = 1: layer 0 -- phase A +
 * its epilogue only (z_prev / dz_prev / w_uo / w_us are not used but must be valid pointers).
 * Rp = 256, Sp = 256, Hp in {128, 192} (BASELINE C2, hps/vqwae.json) run csrc/glu_bwd8.hip since round 6 -- 8 waves x 256 columns, one
 * workgroup per CU, weights AND activation operands through LDS (77.6 against 84 us per launch at C2), bitwise the 4-wave kernel's
 * results; dc_mode bit 2 keeps the 4-wave kernel (the A/B and parity handle of tests/ and tools/time_pair.py). */
int wae_glu_bwd_fused_dc(const wae_glu_bwd_desc* d, const void* dz, int64_t dz_stride, const void* g_next, void* g_out,
                         const void* dskip, const void* z_prev, void* dz_prev, const void* w_x, const void* w_uo,
                         const void* w_us, const void* w_c, float* dc_acc, void* dc_out, int32_t dc_mode, int32_t last,
                         void* stream);

/* ---- all weight-gradient contractions of a step in one launch (csrc/gemm_tn_stream.hip; bf16 operands only) ------
 * Same contraction as wae_gemm_tn_tiles, C[m][n] += alpha * sum_{b,t} P[b,t][m] * Q[b,t+shift][n], cut differently:
 * a job is one 384 x 256 output region over the whole batch; the (job, 32-row time slab) list is cut into one
 * contiguous share per workgroup (segments), so a launch of nwg ~ #CUs workgroups finishes all jobs together and adds
 * each region to C once per (workgroup, job) boundary.  jobs/segs/team_seg are device arrays built by the host
 * (backward.py: StreamTable).  team_size consecutive workgroups form a team that walks one segment list, member m on
 * job (segment job + m) -- the jobs of one layer share operands; team_seg has nteams + 1 prefix offsets into segs;
 * slabs are numbered b * ceil(T/32) + t / 32; a job with m_valid == 0 is skipped; nwg >= nteams * team_size.
 * m_valid <= 384, n_valid <= 256; n_valid % 8 == 0 when ones_col >= 0 (n_valid <= ones_col < 256). */
typedef struct wae_ts_job {
  const void* P;
  const void* Q;
  float* C;
  int64_t p_stride, q_stride, ldc;
  int32_t m_valid, n_valid;
  int32_t shift;
  int32_t ones_col;
  float alpha;
  int32_t pad_;
} wae_ts_job;
typedef struct wae_ts_seg {
  int32_t job, slab_begin, slab_end;
} wae_ts_seg;
/* pace: nteams x 8 int32, ZEROED by the caller before every launch (or NULL / window <= 0: no pacing): members [0, pace_from) of
 * a team (the full-size jobs: the taps) publish the slab position of their next request, and members [pace_from, team_size) (the
 * narrow jobs, which would run ahead) request no more than `window` slabs beyond the slowest of them, so that the operand slabs
 * the jobs of a layer share are fetched from HBM once and found in the XCD's L2 by the others.  Timing only: results do not
 * depend on it, and a wait that does not end switches it off (the launch can never hang). */
int wae_gemm_tn_stream(int32_t dtype /* WAE_BF16 or WAE_F16 */, const wae_ts_job* jobs_dev, const wae_ts_seg* segs_dev,
                       const int32_t* team_seg_dev, int32_t nteams, int32_t team_size, int32_t nwg, int32_t B, int32_t T,
                       int32_t* pace, int32_t window, int32_t pace_from, void* stream);

/* ---- the same launch on a static schedule (csrc/gemm_tn_static.hip; 16-bit operands; round 4) -----------------------
 * Replaces the autograd of modules.py:134-136 (dilated conv: kind TAPS, one job per tap), :141-145 (conv1x1c + the per-clip
 * sums of dz that feed the conv bias and conv1x1g: kind COND) and :157-160 (conv1x1_out AND conv1x1_skip, which contract the
 * same gated activations u: kind OUTSKIP, C0[h][r] += sum u[t][h] Ghat[t][r], C1[h][s] += sum u[t][h] dS[t][s], Cb[r] += sum
 * Ghat[t][r]) for geometries that fit one region: m_valid <= 384 (TAPS / COND) or 192 (OUTSKIP), n0_valid, n1_valid <= 256
 * (COND: <= 64), B <= 32, every operand clip below 2^30 bytes.  Teams, segments and slab numbering as wae_gemm_tn_stream.
 * Rows outside a clip (a tap's causal shift, T % 32 != 0) are zero-filled by the hardware through per-clip buffer
 * descriptors; OUTSKIP's outputs are TRANSPOSED relative to the reference's weight layout (rows = gated channel h).
 * The same kinds carry the rest of the step's weight gradients (one more group of jobs; backward.py: static_head): the head's
 * 1x1 convolutions (wavenet.py:136-141) as TAPS jobs with shift 0 (P = dy, Q0 = h1; P = dh1, Q0 = h0), their biases as COND jobs
 * WITHOUT a Q operand (Q0 = NULL, n0_valid = 0: only the per-clip column sums of P, at ones_col), first_conv (wavenet.py:119-122,
 * 203) as a TAPS job whose P is the one-hot operand of wae_onehot_rows, and -- the last layer's conv1x1_out gradient being dead
 * (wavenet.py:205-207) -- dS as Q0 of that layer's OUTSKIP job, whose Cb then is the skip bias gradient.  A record with
 * m_valid <= 0 is a null job (its team member skips the segment). */
enum { WAE_TQ_TAPS = 0, WAE_TQ_COND = 1, WAE_TQ_OUTSKIP = 2 };
typedef struct wae_tq_job {
  const void* P;
  const void* Q0;
  const void* Q1;
  float* C0;
  float* C1;
  float* Cb;
  int64_t p_stride, q0_stride, q1_stride, ldc0, ldc1;
  int32_t m_valid, n0_valid, n1_valid;
  int32_t shift;
  int32_t ones_col;   /* COND: column of clip 0's sums in C0 (a multiple of 32, >= n0_valid); clip b adds into ones_col + b */
  int32_t kind;
  float alpha;
  int32_t pad_;
} wae_tq_job;
/* stamps: NULL, or (diagnostic builds, -DWAE_TQ_STAMPS) nwg x 16 x 4 int64 zeroed by the caller.
 * pace: nteams uint32, ZEROED by the caller before every launch (or NULL / window <= 0: no pacing): every tap member adds its
 * request progress (in 32-row slabs; a team's share must stay below 1000 slabs) to its 10-bit field, and every TAPS / COND member
 * requests no more than `window` slabs beyond the slowest other tap member, so that the dz half-slabs four jobs of a layer share
 * are fetched from HBM once and found in the XCD's L2 by the others (window_cond: the bound of the COND member; negative = it stays
 * that many slabs behind the slowest tap).  Timing only: results do not depend on it, and a wait that
 * does not end switches it off.
 * max_clip_bytes: the caller's statement of the largest operand clip of the job table, (T + the largest |shift| + 32) rows x the
 * widest p / q row in bytes; refused (WAE_EINVAL) at 2^30 or more, because rows outside a clip are zero-filled by the range check
 * of a per-clip buffer descriptor with 32-bit offsets. */
int wae_gemm_tn_static(int32_t dtype /* WAE_BF16 or WAE_F16 */, const wae_tq_job* jobs_dev, const wae_ts_seg* segs_dev,
                       const int32_t* team_seg_dev, int32_t nteams, int32_t team_size, int32_t nwg, int32_t B, int32_t T,
                       int64_t* stamps, uint32_t* pace, int32_t window, int32_t window_cond, int32_t ntaps, int64_t max_clip_bytes,
                       void* stream);

/* ---- backward of the front end (csrc/frontend_bwd.hip) -----------------------------------------------------
 * upsample stage: dout (B,C,Tin*s), in (B,C,Tin) -> din (B,C,Tin), dw[2s+1] += (atomics).
 * conv block: dpre = dy * [relu ? (y - (residual ? x : 0)) > 0 : 1]; dx (or NULL), dw +=, dbias += (or NULL).
 * VQ: dlat = dquant + loss_scale*2*beta*(x-q)/N; demb[idx] += loss_scale*2*(q-x)/N (straight-through estimator). */
int wae_upsample_stage_bwd(const float* dout, const float* in, const float* w, float* din, float* dw, int32_t B,
                           int32_t C, int32_t Tin, int32_t s, void* stream);
int wae_enc_conv_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                     float* dbias, int32_t B, int32_t Cin, int32_t Tin, int32_t Cout, int32_t k, int32_t stride,
                     int32_t pad, int32_t relu, int32_t residual, void* stream);
int wae_vq_bwd(const float* lat, const float* quant, const int64_t* idx, const float* dquant, float* dlat, float* demb,
               int32_t B, int32_t D, int32_t Tq, float beta, float loss_scale, void* stream);

/* layout helpers: (B,C,T) fp32 <-> (B,T,Cp) dtype */
int wae_to_btc(const float* in, void* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype, void* stream);
/* the same with the shifted loss mask and weight of vqwae_train.py:374-379,764-766 applied on the way: row t is multiplied by
 * `scale` while t + 1 < min(lengths[b], T) (lengths == NULL: T) and zeroed otherwise -- d loss / d y_hat of the masked mean
 * of a per-step loss (the discretized mixture of logistics, mixture.py:26-106) in the layout wae_head_bwd takes as ext_dy */
int wae_to_btc_masked(const float* in, void* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype,
                      const int32_t* lengths, float scale, void* stream);
int wae_from_btc(const void* in, float* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype, void* stream);
/* wae_from_btc with every element multiplied by `scale` (fp16 runs: undoes the loss scale where the conditioning gradient
 * leaves the 16-bit stack for the fp32 front end) */
int wae_from_btc_scaled(const void* in, float* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype, float scale,
                        void* stream);
/* upsample_activation (upsample.py:44-46): the element-wise module the reference puts behind every stage's smoothing FIR --
 * kind 1 nn.ReLU, 2 nn.LeakyReLU(negative_slope = slope), 3 nn.Tanh, 4 nn.Sigmoid.  wae_act_fwd: x <- act(x) in place (fp32);
 * wae_act_bwd: d <- d * act'(.) formed from the activation's OUTPUT y. */
int wae_act_fwd(float* x, int64_t n, int32_t kind, float slope, void* stream);
int wae_act_bwd(const float* y, float* d, int64_t n, int32_t kind, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAE_H */
