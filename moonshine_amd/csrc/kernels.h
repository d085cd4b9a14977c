// Launch wrappers for the hand-written gfx950 kernels.  Every function enqueues on
// `stream` and returns immediately; shapes are in elements.
#pragma once

#include "msh_common.h"

namespace msh {

// RoPE parameters shared by encoder and decoder epilogues
// (interleaved pairs on the first 2*rot_pairs dims of each head,
//  transformers modeling_moonshine.py:196-240).
struct RopeParams {
  const float* cos;  // [max_pos][rot_pairs]
  const float* sin;
  int rot_pairs;
  int head_dim;
  int hidden;  // D = heads * head_dim
};

// ---------------- MFMA-fragment-major ("FM") layouts of the offline decode path ----------------
// The decode GEMMs load their MFMA operands straight from global memory, one 16 x 32 bf16 fragment (16 rows x 64 bytes)
// per wave instruction.  With row-major operands that is sixteen half-used 128-byte lines per instruction, and the
// timeline of the kernels (tools/dec_gemm_timeline.hip) showed the waves spending 1-4.7 us just ISSUING their loads.
// In FM order the 64 lanes of a fragment are contiguous, so every wave-level load is one 1 KiB run (8 full lines):
//   bf16 [R][K]:  element (r, k) at (((r/16) * K/32 + k/32) * 64 + lane) * 8 + k%8,   lane = r%16 + 16 * ((k/8) % 4)
//   fp32 [R][K]:  the lane's 8 values as two float4 halves, each half 1 KiB contiguous per fragment:
//                 ((((r/16) * K/32 + k/32) * 2 + (k/4)%2) * 64 + lane) * 4 + k%4
// Decode weights are repacked to FM at load; the residual stream H (fp32), the attention outputs and the MLP
// activations (bf16) live in FM between the decode kernels.  R is padded to a multiple of 16, K is a multiple of 32.
__host__ __device__ inline long fm16(int r, int k, int ksteps) {
  return ((((long)(r >> 4) * ksteps + (k >> 5)) * 64) + ((r & 15) + 16 * ((k >> 3) & 3))) * 8 + (k & 7);
}
__host__ __device__ inline long fm32(int r, int k, int ksteps) {
  return (((((long)(r >> 4) * ksteps + (k >> 5)) * 2 + ((k >> 2) & 1)) * 64) + ((r & 15) + 16 * ((k >> 3) & 3))) * 4 + (k & 3);
}

// ---------------- tiled MFMA GEMM (large M): C = A[M,K](lda) * W[N,K]^T ----------------
// conv1: out_bf16[M,N] = tanh(acc).  rowsum (optional): [M][column tiles] {sum, sum of squares} of the bf16 values every
// (row, column tile) stored -- the GroupNorm statistics without a pass over the output (groupnorm_stats_rows).  Returns the
// number of column tiles of the kernel it ran (the second dimension of rowsum; at most gemm_max_col_tiles(N)).
int gemm_tanh_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, bf16_t* out, float2* rowsum, hipStream_t s);
inline int gemm_max_col_tiles(int N) { return (N + 63) / 64; }   // the narrowest tile of the tiled family is 64 columns
// conv2 with the GroupNorm folded in (see EpiGnBiasGeluBf16): W = conv2 weight * gamma; table [clips][N] from
// gn_fold_table; stats per clip {mean, rstd}; row m belongs to the clip of stream row m / 2
// kperm = 1: W is stored in the tap-inner k-order (32 channels outer, the 7 taps inner: conv_k_offset, gemm_common.h)
void gemm_gn_bias_gelu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* table, const float2* stats,
                            const int* row_clip, int M, int N, int K, bf16_t* out, int kperm, hipStream_t s);
// table[b][n] = b2[n] - mean_b * rstd_b * s1[n]
void gn_fold_table(const float2* stats, const float* s1, const float* b2, int n_clips, int N, float* table,
                   hipStream_t s);
// out_bf16[M,N] = gelu(acc + bias)
void gemm_bias_gelu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K,
                         bf16_t* out, hipStream_t s);
// conv3: out_f32[M,N] = gelu(acc + bias); kperm as above with 3 taps
void gemm_bias_gelu_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K,
                        float* out, int kperm, hipStream_t s);
// encoder QKV: out_bf16[M,3D] = rope(acc) (q,k parts), position from row_pos[m] (junk rows: pos<0 -> 0)
void gemm_qkv_rope_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_pos,
                        RopeParams rp, bf16_t* out, hipStream_t s);
// H_f32[M,N] += acc (+ bias if non-null)
void gemm_resid_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                    hipStream_t s);
// cross K/V for all decoder layers at once: W = [L*2*D, D]; writes K^T / V^T
// ([L][clip][D][Tk] bf16, zero for t >= T) -- the layout the decode kernel streams.
void gemm_cross_kv(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_clip,
                   const ClipMeta* clips, int D, long layer_stride, bf16_t* KT, bf16_t* VT, hipStream_t s);

// the same as e4m3 bytes: value * qscale[n] (per output column, fixed at load); layer_stride in bytes
void gemm_cross_kv_fp8(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_clip,
                       const ClipMeta* clips, int D, long layer_stride, const float* qscale, uint8_t* KT, uint8_t* VT,
                       hipStream_t s);

// ---------------- decode GEMMs (M = batch rows; k_gemm_dec.hip) ----------------
// "LN" variants take the fp32 residual stream H and fuse LayerNorm (no bias, eps 1e-5) into the A-fragment
// build; the LayerNorm scale gamma must already be folded into W (W' = W * diag(gamma), done at load).
// The dec_* functions below work on FM operands (see above): every W is FM bf16 [N][K], H is FM fp32 [M16][D], the bf16
// activations `A` / `z` are FM [M16][K] (M16 = M rounded up to 16 rows, which the buffers must hold); q, the logits and
// the self-attention cache keep their row-major layouts.
// q/k/v for one decoder layer: q_f32[M,D] (rope), k (rope) / v appended to the self cache
// [M][H][Smax][dh] at position *pos_ptr.
// tile shapes of the decode GEMMs enqueued by THIS host thread from now on: true = fewer, fatter workgroups (a step that
// shares the GPU with other lanes), false = the latency shapes (k_gemm_dec.hip "Throughput shapes")
void dec_gemm_prefer_throughput(bool on);
void dec_gemm_qkv(const float* H, const bf16_t* W, int M, int D, const int* pos_ptr, RopeParams rp, float* q,
                  bf16_t* cacheK, bf16_t* cacheV, int Smax, hipStream_t s);
// out_f32[M,N] = LN(H) * W^T
void dec_gemm_ln_f32(const float* H, const bf16_t* W, int M, int N, int D, float* out, hipStream_t s);
// z_bf16[M,F] = silu(gate) * value of (LN(H) * W^T + bias); W/bias rows interleaved (value_j, gate_j)
void dec_gemm_ln_swiglu(const float* H, const bf16_t* W, const float* bias, int M, int F, int D, bf16_t* z,
                        hipStream_t s);
// H_f32[M,N] += A_bf16[M,K] * W^T (+ bias)
void dec_gemm_resid(const bf16_t* A, const bf16_t* W, const float* bias, int M, int N, int K, float* H, hipStream_t s);
// dy_bf16[M,D] (row-major, for the tiled LM head) = LayerNorm(H) * gamma, H in FM
void dec_final_layernorm(const float* H, const float* gamma, int M, int D, bf16_t* y, hipStream_t s);
// logits_f32[M,V] = LN(H) * E'^T  (E' = tied embedding with the final LayerNorm scale folded in)
void dec_gemm_logits(const float* H, const bf16_t* E, int M, int V, int D, float* logits, hipStream_t s);
// logits_f32[M,N] = A_bf16[M,K] * W^T with the tiled kernel (LM head at batch >= 128, after layernorm_bf16)
void gemm_logits_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s);

// out = act(acc + bias) as bf16 and / or fp32 (either pointer may be null); act 0 none, 1 SiLU, 2 GELU(erf)
void gemm_act(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int act, int M, int N, int K,
              bf16_t* out_bf16, float* out_f32, hipStream_t s);
// z_bf16[M,N/2] = silu(gate) * value of (acc + bias); W / bias rows interleaved (value_j, gate_j)
void gemm_swiglu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, bf16_t* z,
                      hipStream_t s);

// Small-batch (M <= 256) forms of the five GEMMs above on the split-K decode kernel, bf16 input.  Return false
// (nothing launched) when K is not one of the compiled widths; the caller then uses the tiled kernel.
bool small_gemm_act(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int act, int M, int N, int K,
                    bf16_t* out_bf16, float* out_f32, hipStream_t s);
bool small_gemm_qkv_rope_bf16(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, const int* row_pos,
                              RopeParams rp, bf16_t* out, hipStream_t s);
bool small_gemm_swiglu_bf16(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K,
                            bf16_t* z, hipStream_t s);
bool small_gemm_resid_f32(const bf16_t* A, long lda, const bf16_t* W, const float* bias, int M, int N, int K, float* H,
                          hipStream_t s);
bool small_gemm_logits_f32(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* out, hipStream_t s);

// LayerNorm-fused small-batch forms for the streaming decoder's AR steps: H = fp32 residual stream [M][D], Wf = the
// weight with the LayerNorm scale folded in.  small_ln_gemm_stream_qkv writes q to q_out and k / v straight into
// the per-stream self-attention cache (see EpiStreamQkv).  False when D is not a compiled width.
bool small_ln_gemm_stream_qkv(const float* H, const bf16_t* Wf, int M, int D, bf16_t* q_out, bf16_t* cacheK,
                              bf16_t* cacheV, const int* row_slot, const int* row_pos, RopeParams rp, int layer, int L,
                              int Scap, hipStream_t s);
bool small_ln_gemm_bf16(const float* H, const bf16_t* Wf, int M, int N, int D, bf16_t* out, hipStream_t s);
bool small_ln_gemm_swiglu(const float* H, const bf16_t* Wf, const float* bias, int M, int N, int D, bf16_t* z,
                          hipStream_t s);
bool small_ln_gemm_logits(const float* H, const bf16_t* Wf, int M, int N, int D, float* out, hipStream_t s);

// The same AR-step GEMMs on FM operands (see the FM layouts above): Wfm = FM bf16 [N][K] (LayerNorm scale folded in for the
// LN forms), H = FM fp32 [M16][D], A / z = FM bf16 [M16][K]; q_out / out stay row-major (the attention kernels read them).
// False (nothing launched) when the width is not compiled: stream_fm_supported(D, F) is the load-time check.
bool stream_fm_supported(int D, int F);
bool stream_fm_qkv(const float* H, const bf16_t* Wfm, int M, int D, bf16_t* q_out, bf16_t* cacheK, bf16_t* cacheV,
                   const int* row_slot, const int* row_pos, RopeParams rp, int layer, int L, int Scap, hipStream_t s);
bool stream_fm_ln_bf16(const float* H, const bf16_t* Wfm, int M, int N, int D, bf16_t* out, hipStream_t s);
bool stream_fm_ln_swiglu(const float* H, const bf16_t* Wfm, const float* bias, int M, int F, int D, bf16_t* z,
                         hipStream_t s);
bool stream_fm_resid(const bf16_t* A, const bf16_t* Wfm, const float* bias, int M, int N, int K, float* H, hipStream_t s);

// LM head fused with the first stage of the argmax: per row and per 208-column tile the maximum logit and its
// (lowest) column -> pval / pidx [M][gemm_argmax_tiles(N)]; the logits themselves are never written.
int gemm_argmax_tiles(int N);
void gemm_argmax_partials(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, float* pval, int* pidx,
                          hipStream_t s);

// ---------------- encoder MLP block as one kernel (k_mlp.hip) ----------------
// H[R][D] fp32 += fc2(gelu(fc1(LayerNorm(H)) + b1)) + b2, in place; Wp from pack_mlp_weights (LayerNorm scale folded in).
bool mlp_fused_supported(int D, int F);
size_t mlp_packed_elems(int D, int F, bool with_oproj = false);
void pack_mlp_weights(const float* w1, const float* gamma, const float* b1, const float* w2, int D, int F, bf16_t* out,
                      const float* wo = nullptr);
void mlp_fused(float* H, const bf16_t* Wp, const float* b2, int R, int D, int F, hipStream_t s);
// the same with the attention output projection in front: H += AO Wo^T first (AO [R][D] bf16, Wp packed with wo), then the
// MLP block on the result -- replaces gemm_resid_f32 (o-proj) + mlp_fused
void mlp_fused_oproj(float* H, const bf16_t* AO, const bf16_t* Wp, const float* b2, int R, int D, int F, hipStream_t s, bool store_nt = false,
                     bf16_t* yfm = nullptr);
float mlp_microbench(int R, int D, int F, int iters, int abl);
void mlp_fused_host(float* h, int R, int D, int F, const float* w1, const float* gamma, const float* b1, const float* w2,
                    const float* b2, const float* ao = nullptr, const float* wo = nullptr, uint16_t* y_fm = nullptr);

// ---------------- A-stationary panel GEMMs of the encoder (k_panel.hip) ----------------
// Weights packed at load by pack_panel_weights (chunks of 32 output columns, MFMA-fragment order; gamma folded in when given).
size_t panel_packed_elems(int N, int D);
void pack_panel_weights(const float* w, const float* gamma, int N, int D, bf16_t* out);
// encoder QKV: LayerNorm(H) * Wqkv^T; q | k with RoPE -> qk [R][2D] bf16, v -> vt [D][vt_ld] bf16 (transposed).
// Replaces layernorm_bf16 + gemm_qkv_rope_bf16 + the swapped-operand V^T GEMM.  R % 8 == 0.
bool qkv_panel_supported(int D, int head_dim, int rot_pairs);
void qkv_panel(const float* H, const bf16_t* Wp, int R, int D, const int* row_pos, RopeParams rp, bf16_t* qk, bf16_t* vt,
               long vt_ld, hipStream_t s, bool store_nt = false);
// the same with the rows already normalised, bf16, in fragment-major order ([ceil(R / 128) * 128][D]; mlp_fused_oproj yfm)
void qkv_panel_prenorm(const bf16_t* Yfm, const bf16_t* Wp, int R, int D, const int* row_pos, RopeParams rp, bf16_t* qk, bf16_t* vt,
                       long vt_ld, hipStream_t s, bool store_nt = false);
// cross-attention K^T / V^T of all L decoder layers on the same kernel: A = encoder output [R][D] bf16, Wp packed from the
// fused [L * 2 * D][D] weight (no gamma); qscale non-null = e4m3 bytes (value * qscale[column]), layer_stride in bytes then.
// Replaces gemm_cross_kv / gemm_cross_kv_fp8 at large batches.
bool cross_kv_panel_supported(int D);
void cross_kv_panel(const bf16_t* A, const bf16_t* Wp, int R, int D, int L, const int* row_clip, const ClipMeta* clips,
                    long layer_stride, const float* qscale, void* KT, void* VT, hipStream_t s);
float qkv_panel_microbench(int R, int D, int iters, uint16_t* out_qk, uint16_t* out_vt, float* out_h, float* out_w, int* out_pos);

// ---------------- attention ----------------
// encoder self-attention over the packed stream: qk [R,2D] bf16 (q | k, RoPE applied), vt = V^T [D][vt_ld] bf16 (row d,
// stream rows contiguous; vt_ld >= R) -> out [R,D] bf16
void enc_attention(const bf16_t* qk, const bf16_t* vt, long vt_ld, bf16_t* out, const ClipMeta* clips, int n_clips,
                   int max_rows, int D, int heads, hipStream_t s);
float enc_attention_microbench(int variant, int n_clips, int T, int D, int heads, int iters, uint16_t* out_host);
// decode self-attention: q [M,D] f32, cache [M][H][Smax][dh] bf16, keys 0..*pos_ptr -> out [M16,D] bf16 in FM
void dec_self_attention(const float* q, const bf16_t* cacheK, const bf16_t* cacheV, const int* pos_ptr, int M, int D,
                        int heads, int Smax, bf16_t* out, hipStream_t s);
// decode cross-attention: q [M,D] f32, K^T/V^T of one layer -> out [M16,D] bf16 in FM
// kdq / vdq non-null: K^T / V^T are e4m3 bytes and kdq[c] / vdq[c] = 1 / qscale of K / V column c of this layer
void dec_cross_attention(const float* q, const bf16_t* KT, const bf16_t* VT, const ClipMeta* clips, int M, int D,
                         int heads, bf16_t* out, hipStream_t s, const float* kdq = nullptr, const float* vdq = nullptr);

// ---------------- single-clip latency path (k_dec_small.hip; M <= 16) ----------------
// Cross-attention of the projected form split over 64-key slices, one workgroup per (slice, head, clip): LayerNorm + the head's
// query projection + the slice's (max, sum, unnormalised output) -> part [M][heads][ns_max][64] fp32 (slices beyond a clip's
// frames: empty records); Wq_rm = cross-q weight, row-major, LayerNorm scale folded in.  dec_merge_resid: H (FM fp32) += merge(part) Wo^T with the merge as the GEMM's prologue.
bool dec_cross_split_supported(int D, int heads);
int dec_cross_split_slices(int T);
size_t dec_cross_split_part_floats(int M, int heads, int ns_max);
void dec_cross_split(const float* H, const bf16_t* Wq_rm, const bf16_t* KT, const bf16_t* VT, const ClipMeta* clips, int M, int D,
                     int heads, int ns_max, float* part, hipStream_t s);
void dec_merge_resid(const float* part, const bf16_t* Wo_fm, int M, int D, int heads, int ns_max, float* H, hipStream_t s);
// the same attention with one workgroup per (clip, head) walking the slices itself (batches of 5 .. 63 clips, clips of at most
// dec_cross_looped_max_slices() slices): writes the attention output as FM bf16 [M16][D], the A operand of dec_gemm_resid;
// bit-identical to dec_cross_split + dec_merge_resid
int dec_cross_looped_max_slices();
void dec_cross_looped(const float* H, const bf16_t* Wq_rm, const bf16_t* KT, const bf16_t* VT, const ClipMeta* clips, int M, int D,
                      int heads, bf16_t* out_fm, hipStream_t s);
// self-attention over the cache + output projection + residual in one launch (M <= 2; the engine uses it for one clip): H += selfattn(q, cacheK, cacheV) Wo^T
bool dec_self_oproj_supported(int D, int heads, int M);
void dec_self_oproj(const float* q, const bf16_t* cacheK, const bf16_t* cacheV, const int* pos_ptr, const bf16_t* Wo_fm, int M, int D,
                    int heads, int Smax, float* H, hipStream_t s);

// Absorbed form (k_xattn.hip): the attention of all heads of a clip in ONE pass over the encoder output `enc` [R][D] bf16
// (clip b's T frames start at row clips[b].row_start).  qf = the keys-side queries of the heads, LN(h) Wqk^T with the merged
// weight Wqk (softmax scale and log2(e) folded in), as dec_gemm_ln_qt writes them: [M][D / 32][16][32] bf16, value and rounding
// residual in the score product's operand order (gemm_common.h EpiQtFrag) -> ctx [M16][heads * D] bf16 in FM, the A operand
// of dec_gemm_resid with the merged weight Wvo (K = heads * D).  heads == 8, D in {416, 288}.
bool cross_absorbed_supported(int D, int heads);
void dec_gemm_ln_qt(const float* H, const bf16_t* W, int M, int heads, int D, bf16_t* qf, hipStream_t s);
// The same queries from the two factors of Wqk (k_crossq.hip): W1 = scale * Wq * diag(gamma) as FM [heads * 64][D] (a head's
// rows padded to 64 with zeros), W2 = Wk in the order pack_crossq_wk writes; same output order.
bool crossq2_supported(int D, int heads);
void pack_crossq_wk(const float* Wk, int D, int heads, bf16_t* out);   // out: heads * (D / 16) * 1024 elements
void dec_crossq2(const float* H, const bf16_t* W1, const bf16_t* W2, int M, int heads, int D, bf16_t* qf, hipStream_t s);
float crossq2_host(const float* x, const float* wq, const float* wk, int M, int D, float* qt_out, int iters);
// stream_nt: the encoder rows with the non-temporal cache policy -- with one batch on the GPU the rows of a clip are re-read by the
// next layer out of the memory-side cache and the default policy is right; with several batches in flight (lanes) the lanes'
// encoder outputs together do not fit it and only displace the weights every lane shares: +4.7 % overlapped throughput
void dec_cross_absorbed(const bf16_t* qf, const bf16_t* enc, const ClipMeta* clips, int M, int D, int heads, bf16_t* ctx,
                        hipStream_t s, bool stream_nt = false);
float cross_absorbed_host(const float* qt, const float* enc_f32, long R, const int* Ts, const int* row_starts, int M, int D,
                          float* ctx_out, int iters);
int xattn_min_batch();

// same with the query projection fused in: H = fp32 residual stream (FM), Wq = cross-q weight [D,D] ROW-MAJOR with the
// LayerNorm scale folded in (replaces dec_gemm_ln_f32 + dec_cross_attention)
// cross-attention probabilities of the current step (word timestamps): out[clip][layer][head][pos][Tcap] fp32, frames
// [0, T) of the row written; T <= 2048
void dec_cross_attention_probs(const float* q, const bf16_t* KT, const ClipMeta* clips, const int* pos_ptr, int M, int D,
                               int heads, int layers, int layer, int Smax, int Tcap, float* out, hipStream_t s);
void dec_cross_attention_fused_q(const float* H, const bf16_t* Wq, const bf16_t* KT, const bf16_t* VT,
                                 const ClipMeta* clips, int M, int D, int heads, bf16_t* out, hipStream_t s,
                                 const float* kdq = nullptr, const float* vdq = nullptr);

// ---------------- elementwise / reductions ----------------
// pack fp32 clips into the bf16 conv1 input stream (clip b at 384*row_start samples)
void pack_audio(const float* const* clip_ptrs, const ClipMeta* clips, int n_clips, bf16_t* out, long out_elems,
                hipStream_t s);
// row_pos[m] = frame index within its clip (or -1 for padding rows), row_clip[m] = clip id
void build_row_meta(const ClipMeta* clips, int n_clips, int* row_pos, int* row_clip, hipStream_t s);
// GroupNorm(1 group) statistics per clip over the valid [L1, D] block of x1 (bf16) -> stats[b] = {mean, rstd}
void groupnorm_stats(const bf16_t* x1, const ClipMeta* clips, int n_clips, int D, float* partials, float2* stats,
                     hipStream_t s);
// the same statistics from the row sums conv1 left beside its output (gemm_tanh_bf16's rowsum, ntn column tiles per row)
void groupnorm_stats_rows(const float2* rowsum, int ntn, const ClipMeta* clips, int n_clips, int D, float* partials,
                          float2* stats, hipStream_t s);
// y_bf16[r,:] = LayerNorm(x[r,:]) * gamma   (no bias, eps 1e-5); optional fp32 copy
void layernorm_bf16(const float* x, const float* gamma, int rows, int D, bf16_t* y, float* y_f32, hipStream_t s);
// decode bookkeeping after the logits of one step: first-max argmax per row, EOS / budget
// masks, token append, next-input embedding, position advance.
struct DecodeState {
  int32_t* tokens;      // [M][stride]; tokens[b][0] = BOS
  int32_t* counts;      // [M] number of tokens written (incl. BOS)
  int32_t* finished;    // [M]
  int32_t* pos;         // [1] current decode position (= number of steps done)
  int32_t* n_active;    // [1] clips still decoding
  const int32_t* forced;  // nullable [M][stride]: teacher-forced inputs for step i+1 at [b][i+1]
  int32_t stride;
  int32_t eos;
  int32_t ignore_eos;
};
void decode_advance(const float* logits, int M, int V, const ClipMeta* clips, DecodeState st, const float* embed_f32,
                    int D, float* H, hipStream_t s);
// same, from the per-tile (max, index) pairs of gemm_argmax_partials
void decode_advance_partials(const float* pval, const int* pidx, int ntn, int M, const ClipMeta* clips, DecodeState st,
                             const float* embed_f32, int D, float* H, hipStream_t s);
// H[b,:] = embed[BOS]; counters reset
void decode_begin(int M, DecodeState st, int bos, const float* embed_f32, int D, float* H, hipStream_t s);

// tiled-GEMM microbenchmark (ms per launch); see k_gemm.hip
float gemm_microbench(int M, int N, int K, long lda, int cfg, int abl, int iters);

}  // namespace msh
