/* pevae_hip.h -- C ABI of libpevae_hip.so: the MI355X (gfx950) kernels behind the DLS_VAE training step.
 *
 * The reference (oklen/PEVAE) has no FFI of its own: its hot path is Python nn.Modules calling PyTorch
 * (cuDNN / cuBLAS / ATen).  Every entry point below therefore cites the reference call site whose vendor
 * kernel it replaces (file:line into /root/reference).  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked host;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - return 0 on success, <0 on error (PV_ERR_*); pv_last_error() gives the message (thread local);
 *   - no hidden allocations: every scratch buffer is a caller-provided workspace whose size comes from the
 *     matching *_bytes() query, so all entry points can be captured into HIP graphs;
 *   - bf16 buffers are raw uint16 bits; "ld" = leading dimension in ELEMENTS.
 */
#ifndef PEVAE_HIP_H
#define PEVAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_ABI_VERSION 2

const char* pv_last_error(void);
int pv_abi_version(void);
/* host out-params: number of CUs, LDS bytes per CU, gcnArchName copied into arch[arch_len] */
int pv_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ---- bf16 MFMA GEMM core -------------------------------------------------------------------------------
 * C[M,N] = A[M,K] B[N,K]^T (+bias[N], or +bias[M] when bias_per_row != 0); A,B bf16 row-major, K % 64 == 0 (zero padded), lda/ldb % 8 == 0,
 * 16-byte aligned.  Writes fp32 C (C += when accumulate != 0) and/or bf16 Cb.  Replaces the cuBLAS GEMMs inside nn.GRU's input
 * projection (models/enc2dec/base_modules.py:42) and the weight/data-gradient GEMMs of nn.Linear backward
 * for `project` (models/enc2dec/decoders.py:162). */
/* m_live / k_live (device scalars, NULL = none): run-time bounds for a launch whose geometry must stay static (HIP
 * graphs): 128-row tiles starting at or past *m_live exit at once, the K loop stops at round_up(*k_live, 64). */
int pv_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                    int bias_per_row, float* C, int ldc, void* Cb, int ldcb, int accumulate, const int* m_live,
                    const int* k_live, void* stream);

/* The same product stored transposed, fp32: Ct[n * ldct + m] = sum_k A[m][k] B[n][k] + bias[n] (M % 4 == 0, ldct % 4 == 0).
 * A lane's four accumulator rows are 16 contiguous bytes of C^T.  Builds xprojT [3H][tokens] of the recurrence
 * (base_modules.py:42 input-to-hidden product) from A = tokens x E, B = W_ih. */
int pv_gemm_bf16_nt_ct(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, float* Ct,
                       int ldct, void* stream);

/* Split-K form for weight-gradient GEMMs with few output tiles and a long K (dW_ih / dW_hh of the GRU: K = all
 * L*B tokens): `splits` slices of K are computed by separate workgroups into fp32 slabs of `workspace`
 * (pv_gemm_splitk_ws_bytes) and summed in slice order by a second kernel (bitwise reproducible, no atomics);
 * C (+)= the sum.  pv_gemm_splitk_plan picks `splits` so that tiles * splits ~ target_wgs.
 * With m_live (device scalar, rows past it are neither computed nor written) `splits` is an upper limit: the kernel
 * lowers it on the device until live tiles * splits fits the resident workgroups once, so a launch whose row count
 * is only known on the device (the compacted vocabulary dH) still fills the chip; the workspace is sized for the
 * limit. */
int pv_gemm_splitk_plan(int M, int N, int K, int target_wgs, int* splits);   /* host out-param */
size_t pv_gemm_splitk_ws_bytes(int M, int N, int splits);
int pv_gemm_bf16_nt_splitk(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int splits,
                           void* workspace, float* C, int ldc, int accumulate, const int* m_live, const int* k_live,
                           void* stream);
/* C[M][N] (+)= A^T B with BOTH operands K-major: A [K][lda] bf16 (element (m, k) at A[k * lda + m]), B [K][ldb] -- the form
 * of a weight gradient dy^T x over token-major operands (dW_ih = dgi^T x, dW_hh = dgh^T h_prev of BaseRNN's backward,
 * base_modules.py:42; autograd's mm(dy.t(), x)).  The kernel reads the MFMA fragments out of the [k][x] LDS images with
 * transposed reads, so no transposed copy of dy or x is made.  K % 64 == 0; lda >= M and ldb >= N, multiples of 8 (what
 * lies between M / N and the leading dimension is read and must be finite or zero only where it lies below N / M of a
 * tile that is stored); splits > 1: split-K through `workspace` (pv_gemm_splitk_ws_bytes) with the fixed-order fold. */
int pv_gemm_bf16_tn_splitk(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int splits,
                           void* workspace, float* C, int ldc, int accumulate, void* stream);

/* ---- fused vocabulary projection + log-softmax + NLL -----------------------------------------------------
 * Forward: logits = (h W^T + bias) * inv_temp are produced tile-wise by the MFMA GEMM and reduced on the fly
 * to per-token lse[t] and nll[t] = lse - logit[target] (0 where target == ignore_index); the [T,V] log-prob
 * tensor is never written.  Replaces decoders.py:191-193 (Linear + /temperature + log_softmax) fused with
 * criterions.py:55-64 (NLLLoss sum) and criterions.py:133-138 (Perplexity's token mean).
 * h [T][ldh] bf16, W [V][ldw] bf16, H % 64 == 0. */
size_t pv_vocab_nll_fwd_ws_bytes(int T, int V);
int pv_vocab_nll_fwd(const void* h, int ldh, const void* W, int ldw, const float* bias, const int64_t* target,
                     int T, int V, int H, float inv_temp, int64_t ignore_index, void* workspace, float* lse,
                     float* nll, const int* live_rows, void* stream);
/* Backward, stage 1: recompute the logits tile-wise and emit P = (softmax - onehot(target)) * gscale[t] *
 * inv_temp as bf16, both as P [round_up(T,64)][ldp] and P^T [round_up(V,64)][ldpt], plus
 * dbias[v] = sum_t P[t,v].  Stage 2 is two pv_gemm_bf16_nt calls: dH = P (W^T)^T and dW = P^T (h^T)^T.
 * dbias_part: [round_up(T,64)/64][ldp] floats of scratch. */
int pv_vocab_nll_bwd_prob(const void* h, int ldh, const void* W, int ldw, const float* bias,
                          const int64_t* target, const float* lse, const float* gscale, int T, int V, int H,
                          float inv_temp, void* P, int ldp, void* PT, int ldpt, float* dbias_part, float* dbias,
                          const int* live_rows, void* stream);
/* Ignored-token compaction.  Rows whose target is ignore_index add exactly zero to the loss (NLLLoss(ignore_index),
 * criterions.py:38-41) and to every gradient.  pv_vocab_compact: stable partition, perm[new] = old row (live rows first,
 * original order), count[0] = live rows.  pv_vocab_gather: h_c / target_c = rows in perm order.  The vocabulary kernels
 * then run on the compact rows with live_rows = count (tiles past the live rows exit; lse / nll of those rows are 0)
 * and pv_scatter_rows_f32 puts per-row results back: dst[perm[new]] = new < count ? src[new] : 0. */
int pv_vocab_compact(const int64_t* target, int T, int64_t ignore_index, int64_t* perm, int* count, void* stream);
int pv_vocab_gather(const void* h, int ldh, const int64_t* target, const int64_t* perm, int T, int H, void* h_c, int ldc,
                    int64_t* target_c, void* stream);
int pv_scatter_rows_f32(const float* src, long lds, const int64_t* perm, const int* count, int T, int cols, float* dst,
                        long ldd, void* stream);
/* ---- the same group with NOTHING of size [T][V] in HBM, forward and backward (H = 512; pv_vocab_fused_supported) ------
 * Two flash-attention-shaped kernels (csrc/vocab_fused.hip), each two chained MFMA products per 32 x 32 block with the
 * probabilities handed from the first product's accumulator to the second's operand in registers:
 *   pv_vocab_fused_fwd : S^T = W h^T -> online softmax -> O^T += W^T P^T.  Emits lse[t], nll[t] (0 for ignored rows) AND
 *                        U[t][:] = sum_v softmax[t,v] W[v] - W[target[t]] (fp32 [T][512]), the gradient of nll[t] with
 *                        respect to the SCALED logits' input h[t] / inv_temp: dH[t] = dL/dnll[t] * inv_temp * U[t], an
 *                        elementwise pass in backward (pv_scatter_rows_scaled_f32) instead of a GEMM over a stored P.
 *   pv_vocab_fused_dw  : S = h W^T -> P = gscale (softmax - onehot) -> Z += P^T h.  Emits dW (+= when accumulate) and dbias;
 *                        gscale[t] = dL/dnll[t] (0 for ignored rows; pv_vocab_gscale), lse from the forward call.
 * Rows are the compact order of pv_vocab_compact (live rows first; live_rows = its count, NULL = all T rows);
 * target32: the targets as int32 (pv_vocab_gather32).  Replaces decoders.py:191-193 + criterions.py:55-64 and their
 * backward.  workspace: pv_vocab_fused_{fwd,dw}_ws_bytes -- per-workgroup partial results of the stream-K split, folded
 * in a fixed order by a second kernel (bitwise reproducible, no atomics). */
int pv_vocab_fused_supported(int H);
size_t pv_vocab_fused_fwd_ws_bytes(int T, int V);
int pv_vocab_fused_fwd(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, int T, int V,
                       int H, float inv_temp, int64_t ignore_index, void* workspace, float* lse, float* nll, float* U,
                       const int* live_rows, void* stream);
size_t pv_vocab_fused_dw_ws_bytes(int V);
int pv_vocab_fused_dw(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, const float* lse,
                      const float* gscale, int T, int V, int H, float inv_temp, void* workspace, float* dW, int lddw,
                      int accumulate, float* dbias, const int* live_rows, void* stream);
/* pv_vocab_fused_dw on `wgs` workgroups instead of one per CU (0 = one per CU; the stream-K split takes any count) and
 * with dbias accumulated too (accumulate_bias != 0).  Nothing downstream of the vocabulary projection's backward reads dW
 * or dbias before the optimizer, so a caller may run this launch on a second stream beside the latent stack's backward --
 * a chain of small kernels -- and leave that chain the CUs it needs.  No counterpart in the reference. */
int pv_vocab_fused_dw_wgs(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32, const float* lse,
                          const float* gscale, int T, int V, int H, float inv_temp, void* workspace, float* dW, int lddw,
                          int accumulate, float* dbias, int accumulate_bias, const int* live_rows, int wgs, void* stream);
/* The same for rows [v0, v1) of the vocabulary only (v0 a multiple of 128): W, bias, dW and dbias are the WHOLE arrays, the
 * targets keep their global numbers; workspace: pv_vocab_fused_dw_ws_bytes(v1 - v0).  Several ranges may run at the same
 * time on different streams with their own workspaces (ops.VOCAB_DW_BESIDE: half the vocabulary beside each backward
 * recurrence).  No counterpart in the reference. */
int pv_vocab_fused_dw_range(const void* h, int ldh, const void* W, int ldw, const float* bias, const int* target32,
                            const float* lse, const float* gscale, int T, int V, int H, float inv_temp, int v0, int v1,
                            void* workspace, float* dW, int lddw, int accumulate, float* dbias, int accumulate_bias,
                            const int* live_rows, int wgs, void* stream);
/* pv_vocab_gather with the targets also as int32; perm == NULL: identity order (h_c / target_c may be NULL). */
int pv_vocab_gather32(const void* h, int ldh, const int64_t* target, const int64_t* perm, int T, int H, void* h_c, int ldc,
                      int64_t* target_c, int* target32, void* stream);
/* dst[perm[new]][:] = new < *count ? src[new][:] * scale[new] * factor : 0 (perm / count NULL: identity / all rows);
 * cols % 4 == 0 */
int pv_scatter_rows_scaled_f32(const float* src, const float* scale, float factor, const int64_t* perm, const int* count, int T,
                               int cols, float* dst, long ldd, void* stream);
/* The same with the backward of the decoder's output dropout (decoders.py:189) folded into the store: destination row
 * r = i0 * n1p + i1 holds the logical elements i0 * m0 + i1 * m1 + c (pv_dropout_f32's convention); keep decisions from
 * `mask` or regenerated from (rng_state, site).  One pass over dH instead of two. */
int pv_scatter_rows_scaled_drop_f32(const float* src, const float* scale, float factor, const int64_t* perm, const int* count,
                                    int T, int cols, float* dst, long ldd, int n1, int n1p, long m0, long m1, float p,
                                    const void* mask, const void* rng_state, int site, void* stream);
/* Dense [T][V] fp32 log-probs for callers that need the reference's DecoderRNN return value as a tensor
 * (decoders.py:193,292); needs lse from pv_vocab_nll_fwd. */
int pv_vocab_logprob_dense(const void* h, int ldh, const void* W, int ldw, const float* bias, const float* lse,
                           int T, int V, int H, float inv_temp, float* out, void* stream);
/* Greedy decoding without a [T][V] tensor (decoders.py:252-254, scores.topk(1); :144-147,247-248 mask_unk): ids[t] =
 * arg-max over columns != mask_col (mask_col < 0: none; lowest column on ties) of the scaled logits of row t, logp[t] =
 * its log-softmax value.  workspace: pv_vocab_argmax_ws_bytes(T, V). */
size_t pv_vocab_argmax_ws_bytes(int T, int V);
int pv_vocab_argmax(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H, float inv_temp,
                    int mask_col, void* workspace, int64_t* ids, float* logp, void* stream);
/* `sample` decoding without a [T][V] tensor (decoders.py:255-256 -> base_modules.py:54-59: torch.multinomial(exp(log-probs), 1),
 * one draw per row): ids[t] = arg-max over columns != mask_col of (scaled logit - log q) with q ~ Exp(1) -- the sampler
 * torch itself uses for one draw -- and logp[t] = the drawn symbol's log-softmax value.  noise_q (optional, [T][noise_ld]
 * fp32): the q draws (parity tests hand the same draws to the dense path / the reference); NULL: Philox keyed by
 * (rng_state {seed, step}, site, row, column).  workspace: pv_vocab_sample_ws_bytes(T, V). */
size_t pv_vocab_sample_ws_bytes(int T, int V);
int pv_vocab_sample(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H, float inv_temp,
                    int mask_col, const float* noise_q, long noise_ld, const void* rng_state, int site,
                    void* workspace, int64_t* ids, float* logp, void* stream);
/* Beam search's step without a [T][V] tensor (decoders.py:263-270: topk(beam) over cum_sum + the log-softmax rows): per row
 * the k best columns != mask_col of the scaled logits (descending, lowest column first on ties) and their log-softmax
 * values, ids / logp [T][k], 1 <= k <= 16.  Candidates come from the k slabs with the largest per-slab bests of the arg-max
 * epilogue, whose 64 k logits are recomputed in fp32 from the bf16 operands.  workspace: pv_vocab_argmax_ws_bytes(T, V). */
int pv_vocab_topk(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H, float inv_temp,
                  int mask_col, int k, void* workspace, int64_t* ids, float* logp, void* stream);
/* No counterpart in the reference: pv_vocab_argmax for callers that cannot allocate or pass a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  The per-slab partials live in a device array
 * built into the library (one copy per GPU, pv_vocab_argmax_noalloc_max_bytes() = 4 MiB, held until the process ends);
 * no hipMalloc / hipFree / allocating call is made.  Returns PV_ERR_LIMIT (-3) when ceil(V/64) * round_up(T,64) * 16
 * bytes exceed that size.  All calls on one GPU share the array: calls that overlap on one GPU (two streams) race.  The
 * first call on a device looks the array's address up (hipGetSymbolAddress): make it outside stream capture. */
int pv_vocab_argmax_noalloc(const void* h, int ldh, const void* W, int ldw, const float* bias, int T, int V, int H,
                            float inv_temp, int mask_col, int64_t* ids, float* logp, void* stream);
size_t pv_vocab_argmax_noalloc_max_bytes(void);
/* The whole free-running greedy decode in ONE persistent launch: the GEN branch of
 * DecoderRNN.forward with gen_type 'greedy' (decoders.py:236-262) for a one-layer GRU decoder without attention -- per
 * token embedding -> [cat latent] -> GRU step (base_modules.py:42) -> project + log_softmax (:191-193) -> [mask_unk
 * :247-248] -> topk(1) (:252-254) -> next input; all L tokens for every row, no early exit at EOS (as the reference).
 * gx_table [V][3H] fp32 = W_ih[:, :E] emb[v] + b_ih for every vocabulary entry (one pv_gemm_bf16_nt per decode);
 * gx_const (optional) [B][3H] = W_ih[:, E:] latent[b]; whh_bf16 [3H][round_up(H,32)]; wproj_bf16 [V][ldwp] (zero K
 * padding); h0 (optional) [B][H].  Out: seq [B][L] int64 emitted symbols, logp [B][L] their log-probabilities, hn [B][H]
 * the state after the last step.  mask_col < 0: none.  workspace: pv_greedy_decode_ws_bytes(B, H, V, L) bytes, 256-byte
 * aligned; its first word is an error flag (pv_greedy_decode_error: 1 = a bounded inter-workgroup wait timed out).
 * Limits (PV_ERR_LIMIT): B <= 256, H <= 512, 8 (ceil(H/16) - 1) + ceil(B/32) workgroups <= CUs <= 256, and the
 * per-workgroup share of W_proj (round_up(ceil(V / workgroups), 16) rows) + 48 rows of W_hh must fit 160 KiB of LDS
 * (V <= 24 576 at H = 512 on 256 CUs).  The grid must be co-resident: nothing else may occupy the device's CUs. */
size_t pv_greedy_decode_ws_bytes(int B, int H, int V, int L);
int pv_greedy_decode(const float* gx_table, const float* gx_const, const void* whh_bf16, const float* bhh,
                     const void* wproj_bf16, int ldwp, const float* bproj, const float* h0, int B, int H, int V, int L,
                     int sos_id, int mask_col, float inv_temp, void* workspace, int64_t* seq, float* logp, float* hn,
                     void* stream);
int pv_greedy_decode_error(const void* workspace, int* flag);   /* host out-param; synchronises */
/* The same launch with gen_type 'sample' (decoders.py:255-256 -> base_modules.py:54-59, one torch.multinomial draw per row
 * and token): the emitted symbol is arg-max over columns != mask_col of (scaled logit - log q), q ~ Exp(1) from Philox keyed
 * by (rng_state {seed, step}, site0 + token, row, column) -- token by token the draws of pv_vocab_sample(site = site0 +
 * token) --, logp its log-softmax value.  workspace: pv_sample_decode_ws_bytes(B, H, V, L); error flag and limits as above. */
size_t pv_sample_decode_ws_bytes(int B, int H, int V, int L);
int pv_sample_decode(const float* gx_table, const float* gx_const, const void* whh_bf16, const float* bhh,
                     const void* wproj_bf16, int ldwp, const float* bproj, const float* h0, int B, int H, int V, int L,
                     int sos_id, int mask_col, float inv_temp, const void* rng_state, int site0, void* workspace,
                     int64_t* seq, float* logp, float* hn, void* stream);
/* Backward of the dense path: P = (dlogp - exp(logp) * rowsum(dlogp)) * inv_temp as bf16 [T][ldp] (+ dbias). */
int pv_logsoftmax_bwd_bf16(const float* logp, const float* dlogp, int T, int V, float inv_temp, void* P, int ldp,
                           float* dbias, void* stream);

/* ---- persistent GRU recurrence ---------------------------------------------------------------------------
 * torch.nn.GRU semantics (gate order r,z,n) as used at encoders.py:59-62 and decoders.py:182.
 * xprojT [3H][ldx] fp32 = (W_ih x + b_ih) with (time t, batch row b) at column t*Bpad + b (batch row fastest:
 * one 16-byte load feeds the 4 rows a lane owns; from pv_gemm_bf16_nt with the weight as the A operand);
 * whh_bf16 [3H][Hk]; hseq_bf16 [(L+1)][Bpad][Hk] scratch (bf16 states, slot i = state before processing step i);
 * hall [(L+1)][Bpad][H] fp32 states in processing order (slot 0 = h0) -- slots 1..L ARE the layer output
 * (out[b][t] = hall[t+1][b], or hall[L-t][b] when reverse != 0); gsaveT (optional, training) [L][4][H][Bpad];
 * hn [B][H]; sync: pv_gru_sync_bytes(B) bytes, word NG = error flag (1 = a bounded spin timed out).
 * reverse != 0 walks time backwards (bidirectional encoders).  lengths (optional, int32 [B]): packed-sequence
 * semantics of encoders.py:55-65 -- row b keeps its state at time steps t >= lengths[b] (the caller zeroes those
 * output rows).
 * max_wgs <= 0 means "one workgroup per CU of the current device".
 * Limit (PV_ERR_LIMIT): B <= 2720 (the sync buffer's layout: error word and arrival counters in front of the flags). */
int pv_gru_geometry(int B, int H, int* Hk, int* K3, int* Bpad, int* NG);   /* host out-params */
size_t pv_gru_sync_bytes(int B);
/* *flag = 1 if a bounded wait of ANY pv_gru_* / pv_lstm_* launch has timed out since the library was loaded (or since the
 * last call with clear != 0): a process-level word that no launch resets.  A launch's own error word (sync word NG) is
 * zeroed when the launch starts, so under HIP-graph replay only this one survives until somebody looks.  Synchronises the
 * device.  pv_rnn_sync_error_inject sets the word from a one-thread kernel on `stream` (capturable; tests of the reporting
 * path).  No counterpart in the reference (cuDNN's recurrence cannot time out). */
int pv_rnn_sync_error(int* flag, int clear);
int pv_rnn_sync_error_inject(void* stream);
int pv_gru_fwd(const float* xprojT, long ldx, const void* whh_bf16, const float* bhh, const float* h0,
               const int* lengths, int L, int B, int H, int reverse, void* hseq_bf16, float* gsaveT, float* hall,
               float* hn, void* sync, int max_wgs, void* stream);
/* Backward through time.  whhT_bf16 [H][K3] = W_hh^T.  Emits dgi_bf16 / dgh_bf16 [L][Bpad][K3] (gate
 * gradients w.r.t. the input / hidden pre-activations; GEMM operands for dW_ih, dx, dW_hh), dh0 [B][H],
 * db_ih / db_hh [3H].  dbias_part: 2*NG*4*H floats of scratch.  dgx_bf16 (optional, same size as dgh_bf16): scratch for
 * the inter-workgroup exchange of the gate gradients in MFMA-fragment order (faster steps); NULL exchanges through
 * dgh_bf16 itself. */
int pv_gru_bwd(const float* dout, long dout_sb, long dout_st, const float* dhn, const float* gsaveT,
               const float* hall, const void* whhT_bf16, int L, int B, int H, int reverse, void* dgi_bf16,
               void* dgh_bf16, void* dgx_bf16, float* dh0, float* dbias_part, float* db_ih, float* db_hh, void* sync,
               int max_wgs, void* stream);
/* torch.nn.LSTM semantics (gate order i,f,g,o) for rnn_cell == 'lstm' (base_modules.py:42-47, encoders.py:59-62,
 * decoders.py:182): the same persistent kernels with four gates per unit.  xprojT [4H][ldx], whh_bf16 [4H][Hk],
 * gsaveT [L][4][H][Bpad] = i, f, g, o; call [(L+1)][Bpad][H] fp32 cell states (slot 0 = c0); cn [B][H].
 * Backward: whhT_bf16 [H][K4] = W_hh^T (K4 from pv_lstm_geometry); dg_bf16 [L][Bpad][K4] is the single gate-gradient
 * image (operand of dW_ih, dW_hh and dx); db_ih == db_hh [4H].  lengths (optional [B], both calls): rows keep (h, c) at
 * t >= lengths[b] (pack_padded_sequence semantics, encoders.py:55-65); the backward needs one row group per workgroup. */
int pv_lstm_geometry(int B, int H, int* Hk, int* K4, int* Bpad, int* NG);
int pv_lstm_fwd(const float* xprojT, long ldx, const void* whh_bf16, const float* bhh, const float* h0,
                const float* c0, const int* lengths, int L, int B, int H, int reverse, void* hseq_bf16, float* gsaveT,
                float* hall, float* call, float* hn, float* cn, void* sync, int max_wgs, void* stream);
int pv_lstm_bwd(const float* dout, long dout_sb, long dout_st, const float* dhn, const float* dcn,
                const float* gsaveT, const float* hall, const float* call, const void* whhT_bf16, int L, int B, int H,
                int reverse, void* dg_bf16, float* dh0, float* dc0, float* dbias_part, float* db_ih, float* db_hh,
                const int* lengths, void* sync, int max_wgs, void* stream);

/* ---- embedding gather / sorted scatter-add -----------------------------------------------------------------
 * Occurrence r = i0*n1 + i1 reads ids[i0*s0 + i1*s1] and is row i0*n1p + i1 of the output (n1p >= n1).  Replaces nn.Embedding forward at
 * dialog_models.py:758-765 / decoders.py:174 and embedding_dense_backward (atomic on GPU) with a stable
 * radix sort + per-row segmented sum (no atomics, bitwise reproducible). */
int pv_embedding_gather(const float* table, long num_rows, int D, const int64_t* ids, int n, int n1, int n1p,
                        long s0, long s1, float* out, int ldo, void* out_bf16, int ldb, int* oob_counter,
                        void* stream);
size_t pv_embedding_bwd_ws_bytes(int n, int D);
int pv_embedding_bwd(const float* grad_rows, int ldg, const int64_t* ids, int n, int n1, int n1p, long s0, long s1,
                     long num_rows, int D, long padding_idx, int accumulate, float* dtable, void* workspace,
                     size_t ws_bytes, void* stream);
/* The same in two calls, for callers that know the ids long before the gradient (embedding_dense_backward sorts inside the
 * backward pass; the ids of a training step are known when it starts): pv_embedding_sort leaves the sorted (row id,
 * gradient row) pairs in `workspace` (pv_embedding_bwd_ws_bytes(n, D) bytes, kept by the caller in between), pv_embedding_bwd_sorted
 * does the segmented sum from them.  pv_embedding_bwd == the two back to back. */
int pv_embedding_sort(const int64_t* ids, int n, int n1, int n1p, long s0, long s1, long num_rows, int D, void* workspace,
                      size_t ws_bytes, void* stream);
int pv_embedding_bwd_sorted(const float* grad_rows, int ldg, int n, long num_rows, int D, long padding_idx, int accumulate,
                            float* dtable, void* workspace, size_t ws_bytes, void* stream);
/* No counterpart in the reference (it is single-process): the data-parallel exchange of a table whose local gradient is
 * non-zero only in the n rows looked up this step (user / item tables, dialog_models.py:758-759), as (ids, rows) instead
 * of an all-reduce of the whole table.  pv_sparse_pack: occurrence r sends its id (as fp32, exact: num_rows <= 2^24) and
 * the gradient row of its id if it is the first occurrence of that id, zeros otherwise.  pv_sparse_merge: `gathered`
 * holds every rank's [ids n][rows n x D] block at rank_stride floats from each other (one all-gather); each id's rows are
 * summed in (rank, occurrence) order -- identical bits on every rank -- and overwrite that row of table_grad. */
int pv_sparse_pack(const float* table_grad, long num_rows, int D, const int64_t* ids, int n, float* out_ids,
                   float* out_rows, void* stream);
int pv_sparse_merge(const float* gathered, int world, long rank_stride, int n, int D, long num_rows, float* table_grad,
                    void* stream);

/* ---- layout helpers ------------------------------------------------------------------------------------------ */
/* dst[(i0*n1p + i1)*ldd + c] = bf16(src[i0*s0 + i1*s1 + c*sc]) for c < C, 0 for C <= c < ldd */
int pv_cast_pack_bf16(const float* src, int n0, int n1, int n1p, int C, long s0, long s1, long sc, void* dst,
                      int ldd, void* stream);
int pv_transpose_to_bf16(const void* src, int src_is_f32, int R, int C, long lds, void* dst, long ldd, int Rpad,
                         void* stream);
int pv_colsum_f32(const float* X, int R, long ld, int n, float* out, int accumulate, void* stream);
/* out2[0] = sum of x[i] over i with mask[i] != ignore (all i when mask == NULL), out2[1] = their count */
int pv_masked_sum(const float* x, const int64_t* mask, int64_t ignore, long n, float* out2, void* stream);
/* Its backward, dx[i] = g[0] where mask[i] != ignore (everywhere when mask == NULL), and the row scale of the vocabulary
 * backward, gscale[t] = dnll[perm ? perm[t] : t] * (target[t] != ignore): one launch each for what was a chain of
 * fill / compare / cast / gather / multiply kernels behind criterions.py:55-64. */
int pv_masked_sum_bwd(const int64_t* mask, int64_t ignore, const float* g, long n, float* dx, void* stream);
/* No counterpart in the reference (autograd sums the gradients of a tensor's consumers there, one launch per addend): the
 * gradient sums at the fan-out points of the latent stack -- a pyramid activation feeds the next Block, a DLS_Auto level
 * and the InfoNCE term (dialog_models.py:3284-3307) -- for up to PV_SUM_GROUP tensors in one launch:
 * dst[i][0..n[i]) = src[i][0] + ... + src[i][PV_SUM_SRC-1] (NULL addends skipped; added left to right); n[i] % 4 == 0,
 * 16-byte aligned pointers. */
#define PV_SUM_GROUP 8
#define PV_SUM_SRC 4
typedef struct pv_sum_group_desc {
  float* dst[PV_SUM_GROUP];
  const float* src[PV_SUM_GROUP][PV_SUM_SRC];
  long n[PV_SUM_GROUP];
  int count;
} pv_sum_group_desc;
int pv_sum_group(const pv_sum_group_desc* g, void* stream);
/* NLLEntropy(avg_type 'seq') and Perplexity of one step (criterions.py:64 loss / batch_size; :137 exp(loss / n_words)) from
 * sums2 = {sum of the token NLL, number of live tokens} (pv_masked_sum): out2 = {total / batch, exp(total / count)} in one
 * launch; backward: dsums2 = {g_nll / batch + g_ppl * ppl / count, 0} (either gradient may be NULL). */
int pv_nll_head_fwd(const float* sums2, int batch, float* out2, void* stream);
int pv_nll_head_bwd(const float* sums2, const float* out2, const float* g_nll, const float* g_ppl, int batch, float* dsums2,
                    void* stream);
int pv_vocab_gscale(const float* dnll, const int64_t* perm, const int64_t* target, int64_t ignore, int T, float* gscale,
                    void* stream);
/* dense-input NLL (criterions.py:57-63 on a materialised log-prob tensor): out[t] = -logp[t][target[t]] */
int pv_pick_nll_fwd(const float* logp, const int64_t* target, int64_t ignore, int T, int V, float* out,
                    void* stream);
int pv_pick_nll_bwd(const float* dout, const int64_t* target, int64_t ignore, int T, int V, float* dlogp,
                    void* stream);

/* ---- latent connectors: fused reparameterisation + KL, Gumbel-softmax, categorical KL ------------------------
 * mode 0: NormKLLoss (criterions.py:145-155, logvar = log sigma^2); mode 1: gaussian_analytical_kl
 * (dialog_models.py:3215-3217, inputs log sigma).  z = mu_q + exp(lv_q) * eps (dialog_models.py:191-203,
 * :323-325, :3206-3207); eps == NULL gives z = mu_q (eval). */
int pv_reparam_kl_fwd(const float* mu_q, const float* lv_q, const float* mu_p, const float* lv_p, const float* eps,
                      int B, int Z, int mode, float* z, float* kl_row, void* stream);
int pv_reparam_kl_bwd(const float* mu_q, const float* lv_q, const float* mu_p, const float* lv_p, const float* eps,
                      const float* dz, const float* dkl_row, int B, int Z, int mode, float* dmu_q, float* dlv_q,
                      float* dmu_p, float* dlv_p, void* stream);
/* nn_lib.py:226-265 with the uniform draw `u` supplied by the caller. */
int pv_gumbel_softmax_fwd(const float* logits, const float* u, int rows, int k, float tau, int hard, float* y_soft,
                          float* y, int64_t* ids, void* stream);
int pv_softmax_bwd_rows(const float* y_soft, const float* dy, int rows, int k, float scale, float* dlogits,
                        void* stream);
/* criterions.py:161-175 (mode 0) and :200-211 (mode 1): per-row sums. */
int pv_cat_rows_fwd(const float* log_q, const float* log_p, long log_p_numel, int rows, int k, int mode, float* out,
                    void* stream);
int pv_cat_rows_bwd(const float* log_q, const float* log_p, long log_p_numel, const float* drow, int rows, int k,
                    int mode, float* dlog_q, float* dlog_p, void* stream);

/* ---- small exact-fp32 strided GEMM (Hidden2Gaussian nn_lib.py:198-199, LinearConnector nn_lib.py:143, Block
 * dialog_models.py:258-263, attention bmm decoders.py:85-107) ------------------------------------------------- */
int pv_linear_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias,
                  float* C, long ldc, int M, int N, int K, float alpha, const float* alpha_dev, int accumulate,
                  int batch, long bsa, long bsb, long bsc, void* stream);   /* alpha_dev: optional device scalar factor */

/* ---- DLS latent pyramid pieces (dialog_models.py:253-271 Block, :1835-1870 DlsRec/DlsPri, :3161-3352 DLS_Auto) ------
 * LayerNorm over a residual sum (Block, :269: layer_norm(x + xhat)); y may be NULL. */
int pv_layernorm_fwd(const float* x, const float* y, const float* gamma, const float* beta, int rows, int D, float eps,
                     float* out, float* mean, float* rstd, void* stream);
int pv_layernorm_bwd(const float* x, const float* y, const float* gamma, const float* dout, const float* mean,
                     const float* rstd, int rows, int D, float* ds, float* dgamma, float* dbeta, int accumulate_params,
                     void* stream);
/* torch.nn.utils.spectral_norm of the Block linears (:258-263): one power iteration (do_iter != 0: u, v updated in
 * place) and inv_sigma[0] = 1 / (u . W v).  Backward correction of the weight gradient:
 * dW -= <dy, y - bias> * inv_sigma * u v^T  (dW must already hold (dy^T x) * inv_sigma); scratch1: one float. */
int pv_spectral_norm(const float* W, float* u, float* v, int n_out, int n_in, float eps, int do_iter, float* inv_sigma,
                     void* stream);
/* Multi-tensor form: `descs` is a device array of `count` records {const float* W; float* u; float* v;
 * float* inv_sigma; int64 n_out; int64 n_in; float* Wt} (56 bytes each); one workgroup per weight, one launch per step.
 * transpose != 0: records with Wt != NULL also get Wt [n_in][n_out] = W^T (the weight layout pv_block_fwd streams).
 * workspace (pv_spectral_norm_multi_ws_bytes, or NULL): with it the iteration is split at its two norms into three
 * launches whose workgroups own 64-column / 64-row slabs (the whole chip reads the weights); without it one workgroup
 * walks each weight. */
size_t pv_spectral_norm_multi_ws_bytes(int count, int max_out);
int pv_spectral_norm_multi(const void* descs, int count, int max_out, int max_in, float eps, int do_iter, int transpose,
                           void* workspace, void* stream);
/* One pass over dy: db[n] (+)= sum_m dy (db may be NULL) and cpart[j] = sum over the j-th 64-column block of
 * dy * (y - bias); sum_j cpart[j] = <dL/dW_eff, W_eff>, consumed by pv_linear_f32_fused's rank-1 epilogue. */
int pv_spectral_norm_reduce(const float* dy, const float* y, const float* bias, int M, int N, float* db,
                            int accumulate_db, float* cpart, void* stream);
int pv_linear_f32_fused(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias,
                        float* C, long ldc, int M, int N, int K, float alpha, const float* alpha_dev, int accumulate,
                        float* gelu_out, const float* gelu_grad, const float* r1_c, int r1_nc,
                        const float* r1_u, const float* r1_v, void* stream);
int pv_spectral_norm_bwd(const float* dy, const float* y, const float* bias, long M, int n_out, int n_in, const float* u,
                         const float* v, const float* inv_sigma, float* scratch1, float* dW, void* stream);
/* A whole Block (:253-271) per launch: a workgroup owns 16 batch rows and walks gelu -> c1 -> gelu -> c2 -> gelu ->
 * c3 -> gelu -> c4 -> +x -> LayerNorm -> translate with the activations in LDS (every step is row-independent once
 * 1/sigma is known).  The forward pass streams the TRANSPOSED weights Wt [in][out] written by pv_spectral_norm_multi, the
 * backward pass weight_orig [out][in]: every weight access of both passes is a contiguous 256-byte run.  Widths must be
 * multiples of 16 and <= 512.  pv_block_layout gives the per-row float counts of `save` (activations kept for the
 * backward pass), `dY` (gradients of the four inner linears) and of one workgroup's row in `ppart`.  pv_block_bwd writes
 * dx and (accumulate != 0: adds to) every parameter gradient, including the spectral-norm rank-1 term of the weights. */
int pv_block_layout(int Din, int mid, int Dout, int* save_width, int* grad_width, int* part_width);
#define PV_BLOCK_GROUP_MAX 3
typedef struct pv_block_call {      /* one Block of a launch; arrays are in the order c1, c2, c3, c4, translate */
  int M, Din, mid, Dout;
  float ln_eps;
  int accumulate;                   /* backward: add to the parameter gradients instead of overwriting them */
  const float* W[5];                /* weight_orig [out][in]           (backward) */
  const float* Wt[5];               /* W^T [in][out] of this step      (forward)  */
  const float* bias[5];
  const float* inv_sigma[5];        /* device scalars */
  const float* u[5];                /* power-iteration vectors of this step (backward) */
  const float* v[5];
  const float* gamma;
  const float* beta;
  const float* x;                   /* [M][ldx] */
  long ldx;
  float* save;                      /* [M][save_width]; written by forward, read by backward */
  float* mean;                      /* [M] */
  float* rstd;                      /* [M] */
  float* out;                       /* [M][ldo] */
  long ldo;
  const float* dout;                /* backward: [M][lddo] */
  long lddo;
  float* dx;                        /* [M][lddx] */
  long lddx;
  float* dY;                        /* scratch [M][grad_width] */
  float* cpart;                     /* scratch 5 * 4 * cdiv(M,16) floats (one partial per member of a 4-workgroup cluster) */
  float* ppart;                     /* scratch cdiv(M,4) * part_width floats (one row per 4-row slab) */
  float* dW[5];
  float* db[5];
  float* dgamma;
  float* dbeta;
} pv_block_call;
/* Up to PV_BLOCK_GROUP_MAX INDEPENDENT Blocks per launch (the two pyramids' Blocks of a level, or the three Blocks of a
 * DLS_Auto level, run side by side).  Default kernels (round 3): one workgroup per FOUR batch rows walks the whole Block
 * alone (v_mfma_f32_4x4x1_16b_f32, exact fp32; cdiv(M,4) workgroups per Block, no hand-offs).  With pv_block_rows(16, ..)
 * the round-2 kernels run instead: 16-row slabs, and Blocks 512 wide walked by clusters of 4 workgroups per slab that
 * split every layer's columns and hand their slices to each other through `save` (forward) / `dx` and `dY` (backward),
 * with bounded waits (pv_block_sync_error reports one that gave up); launches of those on ONE device must not overlap
 * each other (each kind shares the device's hand-off counters); stream order, as every caller here uses, is enough. */
int pv_block_fwd(const pv_block_call* calls, int n, void* stream);
int pv_block_bwd(const pv_block_call* calls, int n, void* stream);
/* pv_block_bwd in two launches: phase 1 = the dx chain (block_bwd_dx_kernel: dL/dx, dY, the per-workgroup partials),
 * phase 2 = everything nothing downstream waits for (the weight-gradient GEMMs and the bias / LayerNorm parameter
 * reductions, which read what phase 1 wrote); phase 3 = both = pv_block_bwd.  A caller may issue phase 2 later and on
 * another stream, ordered after phase 1, so that it runs beside the next Block's dx chain.  No counterpart in the
 * reference (autograd orders its per-linear backward nodes itself). */
int pv_block_bwd_phase(const pv_block_call* calls, int n, int phase, void* stream);
/* *flag = 1 if an inter-workgroup wait of pv_block_fwd has timed out since the library was loaded (synchronises the
 * device; tests and debugging -- a timed-out launch has produced garbage).  No counterpart in the reference. */
int pv_block_sync_error(int* flag);
/* Which kernels pv_block_fwd / pv_block_bwd* launch: rows = 4 (default) or 16, wide != 0 (default) = the 4-row kernels also
 * for launches holding a 512-wide Block (0: those keep the 16-row clusters); a negative argument leaves that setting as
 * it is (environment at load: PEVAE_BLOCK_ROWS, PEVAE_BLOCK_ROWS_WIDE).  Returns the rows value now in force.  The forward
 * pass, the dx chain and the parameter-gradient launch of one step must run under one setting (the number of
 * per-workgroup partial rows in cpart / ppart follows it).  No counterpart in the reference (tests and A/B timing). */
int pv_block_rows(int rows, int wide);
/* One level of DLS_Auto's train branch (:3285-3311) as two elementwise launches.  Q / Pr [B][2n]: outputs of dis_layer /
 * dis_layer2 (mean | log sigma halves).  ls = mean + exp(logsigma) * eps + a2 (:3299-3300), out = lay + ls (:3310),
 * kl[0] = (ew + cw) * mean_b sum_d KL(prior || posterior) (:3305-3306; the backward gives the prior's gradient the
 * weight ew and the posterior's cw, as the two detached-direction terms do).  pv_dls_inputs_*: q_in = [a1 + a2 | ls],
 * p_in = [a2 | lt] (:3289-3297) and the split of their gradients.  kl_part: 256 floats of scratch (per-workgroup KL
 * partials, added in index order). */
int pv_dls_level_fwd(const float* Q, const float* Pr, const float* a2, const float* eps, const float* lay, int B, int n,
                     float ew, float cw, float* ls, float* out, float* kl, float* kl_part, void* stream);
int pv_dls_level_bwd(const float* Q, const float* Pr, const float* eps, const float* d_ls, const float* d_out,
                     const float* g_kl, int B, int n, float ew, float cw, float* dQ, float* dPr, float* da2, float* dlay,
                     void* stream);
/* pv_dls_level_fwd / _bwd that also form the NEXT level's Block inputs (:3289-3297) in the same launch: q_in [B][wn + n] =
 * [a1n + a2n | ls], p_in = [a2n | t_mean] (t_mean = the first n columns of Pr); backward takes dq_in / dp_in (either may be
 * NULL) and also emits da1n, da2n [B][wn].  q_in == NULL (da1n == NULL): exactly pv_dls_level_fwd (_bwd). */
int pv_dls_level_next_fwd(const float* Q, const float* Pr, const float* a2, const float* eps, const float* lay, int B, int n,
                          float ew, float cw, float* ls, float* out, float* kl, float* kl_part, const float* a1n,
                          const float* a2n, int wn, float* q_in, float* p_in, void* stream);
/* ... and kl[0] = kl_add[0] + the level's KL term (kl_add != NULL): the running loss is carried through the levels without an
 * add launch per level (the gradient of kl_add is the gradient of kl). */
int pv_dls_level_next_add_fwd(const float* Q, const float* Pr, const float* a2, const float* eps, const float* lay, int B, int n,
                              float ew, float cw, float* ls, float* out, float* kl, float* kl_part, const float* a1n,
                              const float* a2n, int wn, float* q_in, float* p_in, const float* kl_add, void* stream);
int pv_dls_level_next_bwd(const float* Q, const float* Pr, const float* eps, const float* d_ls, const float* d_out,
                          const float* g_kl, int B, int n, float ew, float cw, float* dQ, float* dPr, float* da2, float* dlay,
                          const float* dq_in, const float* dp_in, int wn, float* da1n, float* da2n, void* stream);
int pv_dls_inputs_fwd(const float* a1, const float* a2, const float* ls, const float* lt, int B, int w, int e, float* q_in,
                      float* p_in, void* stream);
int pv_dls_inputs_bwd(const float* dq, const float* dp, int B, int w, int e, float* da1, float* da2, float* dls, float* dlt,
                      void* stream);
/* DLS_Auto.cons_loss (:3266-3272): row L2 normalisation and the symmetric InfoNCE over a [B][B] score matrix with
 * the diagonal as targets: loss = (CE(log_softmax rows) + CE(log_softmax columns)) / 2. */
/* DLS_Auto.cons_loss (dialog_models.py:3266-3272) for n <= 7 (v1, v2) pairs of [B, D[p]] rows at once: every pair of a
 * training step depends only on the pyramids' activations, so the model's five calls (:3284, :3307 per level) become one
 * launch per stage.  loss[0] = sum over pairs of (CE(log_softmax(alpha n1 n2^T, rows), diag) + CE(.., columns))/2 with
 * n = rows / |rows|; v1 / v2 / dv1 / dv2: HOST arrays of n device pointers; ws: pv_cons_loss_ws_floats floats, kept by
 * the caller from the forward to the backward call (it holds the normalised rows, scores and log-sum-exps); gout [1]. */
size_t pv_cons_loss_ws_floats(const int* D, int n, int B);
int pv_cons_loss_fwd(const float* const* v1, const float* const* v2, const int* D, int n, int B, float alpha, float* ws,
                     float* loss, void* stream);
int pv_cons_loss_bwd(const float* gout, const int* D, int n, int B, float* ws, float* const* dv1, float* const* dv2,
                     void* stream);
int pv_l2norm_rows_fwd(const float* x, int rows, int D, float* a, float* nrm, void* stream);
int pv_l2norm_rows_bwd(const float* a, const float* nrm, const float* da, int rows, int D, float* dx, void* stream);
int pv_infonce_fwd(const float* S, int B, float* lse_r, float* lse_c, float* loss, void* stream);
int pv_infonce_bwd(const float* S, const float* lse_r, const float* lse_c, const float* gout, int B, float* dS,
                   void* stream);

/* ---- decoder attention (decoders.py:89-118): fused additive scores, row softmax, tanh / GELU ---------------------
 * score[b,i,j] = qw . tanh(k[b,j,:] + q[b,i,:]) + qb without the [B,Lo,Li,D] intermediate of decoders.py:92-94.
 * dqw_part: B*Lo*D floats of scratch. */
int pv_additive_scores_fwd(const float* q, const float* k, const float* qw, const float* qb, int B, int Lo, int Li,
                           int D, float* score, void* stream);
int pv_additive_scores_bwd(const float* q, const float* k, const float* qw, const float* dscore, int B, int Lo,
                           int Li, int D, float* dq, float* dk, float* dqw_part, float* dqw, void* stream);
int pv_softmax_rows(const float* x, int rows, int n, float* y, void* stream);
int pv_softmax_rows_bwd(const float* y, const float* dy, int rows, int n, float* dx, void* stream);
/* op 0: tanh(a); 1: b*(1-a^2) (a = tanh output); 2: gelu(a) (erf form, dialog_models.py:266-269); 3: b*gelu'(a) */
int pv_unary_f32(const float* a, const float* b, long n, int op, float* out, void* stream);

/* ---- dropout (nn.Dropout of encoders.py:54 and decoders.py:175,189) -------------------------------------------------
 * y = x * keep / (1 - p) over a strided fp32 view [n0][n1][C] (element (i0, i1, c) at src[i0*s0 + i1*s1 + c*sc]).  keep
 * comes from `mask` (uint8, 1 = keep) when given, else from Philox4x32-10 keyed by (seed, step) of `rng_state`
 * (pv_rng_state_bytes() of device memory: {uint64 seed; uint64 step}), the call-site id `site` and the element's LOGICAL
 * index i0*m0 + i1*m1 + c -- the same call on the upstream gradient is the backward pass (decisions are regenerated, no
 * mask is stored), and any layout of the same logical tensor sees the same decisions.  pv_rng_advance: step += 1 (one
 * thread; once per training step, so a replayed HIP graph draws fresh masks).  pv_dropout_pack_bf16 writes the result as
 * the zero-padded bf16 GEMM operand rows i0*n1p + i1 of ldd elements (decoder states -> vocabulary projection). */
size_t pv_rng_state_bytes(void);
int pv_rng_advance(void* state, void* stream);
int pv_dropout_f32(const float* src, int n0, int n1, int C, long s0, long s1, long sc, long m0, long m1, float p,
                   const void* mask, const void* rng_state, int site, float* dst, long d0, long d1, void* stream);
int pv_dropout_pack_bf16(const float* src, int n0, int n1, int n1p, int C, long s0, long s1, long sc, long m0, long m1,
                         float p, const void* mask, const void* rng_state, int site, void* dst, int ldd, void* stream);

/* ---- clip + skip guards + Adam over one flat buffer (main_pretrain.py:207-225,268-269;
 * dialog_models.py:122-131) ------------------------------------------------------------------------------------ */
/* state: pv_optim_state_bytes() of device memory, zeroed once; its first 48 bytes are {float avg_norm; int have_avg,
 * jump_cnt, batch_cnt, adam_step; float last_norm, clip_coef; int skip; float lr_eff, bc1, bc2; int skipped_total},
 * followed at byte 48 by {float lr_base}: with lr < 0 the base learning rate is read from there (the host writes it, e.g.
 * at a plateau decay, main_pretrain.py:343-348), so guards + clip + Adam can be replayed from a HIP graph whose
 * kernel arguments are frozen.
 * grad_scale: the gradient is grad_scale * grads (data parallel: grads = SUM over ranks, grad_scale = 1/world; norm,
 * clip and update use the scaled values without another pass).  weight_decay: decoupled decay of torch.optim.AdamW
 * (the freeze-phase optimizer, dialog_models.py:834: p *= 1 - lr * wd before the update); 0 = torch.optim.Adam. */
size_t pv_optim_state_bytes(void);
int pv_grad_guard(const float* grads, long n, void* state, float max_norm, float lr, int warmup_steps, float beta1,
                  float beta2, int guards, float grad_scale, void* stream);
int pv_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, const void* state,
                 float beta1, float beta2, float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PEVAE_HIP_H */
