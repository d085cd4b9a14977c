/* Development hooks of libmoonshine: kernel-alone test entry points and microbenchmarks.
 *
 * NOT part of the product library: `python -m moonshine_amd.build` links moonshine_amd/lib/libmoonshine.so (the drop-in: the
 * whole of moonshine-c-api.h + moonshine_hip.h, no msh_test_* symbol) and moonshine_amd/lib/libmoonshine_dev.so (the same
 * objects + csrc/dev_hooks.cpp), which the kernel-alone tests (tests/test_gpu_mlp.py, test_gpu_panel.py, test_gpu_xattn.py,
 * test_gpu_guard.py) and the microbenchmark tools load.  No reference counterpart. */
#ifndef MOONSHINE_HIP_DEV_H
#define MOONSHINE_HIP_DEV_H

#include "moonshine_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Developer hook: self-test of the library's device allocator (odd sizes, pageable copies, interior slices); 0 = ok.
 * Meant for MSH_GUARD_ALLOC=1 (tools/gpu_guard.sh), where every buffer ends on an unmapped page. */
MSH_EXPORT int32_t msh_test_device_alloc(void);
/* Developer hook: ms per launch of one tiled-GEMM configuration on synthetic operands (tools/gemm_microbench.py). */
MSH_EXPORT float msh_test_gemm_microbench(int32_t M, int32_t N, int32_t K, int64_t lda, int32_t cfg, int32_t abl,
                                          int32_t iters);

/* Developer hook: ms per launch of the fused encoder MLP kernel (LayerNorm + fc1 + GELU + fc2 + residual, k_mlp.hip) on R
 * rows of uniform random data (tools/mlp_microbench.py); abl = 0, or an ablation of k_mlp.hip (garbage results). */
MSH_EXPORT float msh_test_mlp_microbench(int32_t R, int32_t D, int32_t F, int32_t iters, int32_t abl);

/* Test hook: the fused encoder MLP kernel alone -- h [R][D] (host, in place) += fc2(gelu(fc1(LayerNorm(h) * gamma) + b1)) + b2
 * with w1 [F][D], w2 [D][F] fp32 (rounded to bf16 inside, as at load).  D in {64, 288, 416}, F % 32 == 0. */
MSH_EXPORT int32_t msh_test_mlp_run(float* h, int32_t R, int32_t D, int32_t F, const float* w1, const float* gamma,
                                    const float* b1, const float* w2, const float* b2);

/* Test hook: the same kernel with the attention output projection in front (h += ao wo^T first, ao [R][D], wo [D][D] fp32,
 * both rounded to bf16 inside), as the encoder runs it from 16 k rows on. */
MSH_EXPORT int32_t msh_test_mlp_oproj_run(float* h, int32_t R, int32_t D, int32_t F, const float* w1, const float* gamma,
                                          const float* b1, const float* w2, const float* b2, const float* ao, const float* wo);

/* Test hook: msh_test_mlp_oproj_run that also returns what the kernel hands the next layer's QKV projection (round 6):
 * y_fm [ceil(R / 128) * 128][D] bf16 bit patterns = LayerNorm (no scale, eps 1e-5) of the new rows in fragment-major order --
 * element e of lane l of k-step s of 32-row block w (at ((w * D / 16 + s) * 64 + l) * 8 + e) is row 32 w + (l & 31), column
 * 16 s + 8 (l >> 5) + e. */
MSH_EXPORT int32_t msh_test_mlp_oproj_y_run(float* h, int32_t R, int32_t D, int32_t F, const float* w1, const float* gamma,
                                            const float* b1, const float* w2, const float* b2, const float* ao, const float* wo,
                                            uint16_t* y_fm);

/* Developer / test hook: the encoder QKV panel kernel (LayerNorm + q | k with RoPE + V transposed, k_panel.hip) on R rows
 * (R % 8 == 0) of synthetic data at width D (416 or 288); returns ms per launch (< 0 on error).  Non-null outputs receive the
 * last launch's results as bf16 bit patterns (qk [R][2D], vt [D][R]) and the inputs used (h [R][D], w [3D][D] fp32, pos [R],
 * pos < 0 = padding row) so that a test can recompute them (tests/test_gpu_panel.py). */
MSH_EXPORT float msh_test_qkv_panel(int32_t R, int32_t D, int32_t iters, uint16_t* out_qk, uint16_t* out_vt, float* out_h,
                                    float* out_w, int32_t* out_pos);

/* Test / developer hook: the absorbed cross-attention kernel alone.  M clips, clip b = Ts[b] rows of `enc` [R][D] (fp32,
 * rounded to bf16 inside) from row row_starts[b]; qt [M][8 * D] fp32 (scores = qt_h . enc[t], already in the exp2 domain);
 * ctx_out [M][8 * D] fp32 receives the kernel's bf16 output (softmax_t(qt_h . enc[t]) weighted sum of the rows, per head).
 * D = 416 or 288.  Returns ms per launch over `iters` launches (0 = a single untimed launch), < 0 on error. */
MSH_EXPORT float msh_test_cross_absorbed(const float* qt, const float* enc, int64_t R, const int32_t* Ts,
                                         const int32_t* row_starts, int32_t M, int32_t D, float* ctx_out, int32_t iters);

/* Test / developer hook: the two-stage query kernel of the absorbed form alone (k_crossq.hip).  x [M][D] fp32 (residual
 * stream rows), wq [D][D] = the scaled, LayerNorm-folded query projection with rows (head, j), wk [D][D] the key projection
 * with rows (head, j); qt_out [M][8 * D] fp32 receives qt_h = Wk_h^T (wq_h LN(x)) per head (value + rounding residual of the
 * kernel's split-bf16 output).  D = 416 or 288.  Returns ms per launch over `iters` launches (0 = one untimed launch), < 0 on error. */
MSH_EXPORT float msh_test_crossq2(const float* x, const float* wq, const float* wk, int32_t M, int32_t D, float* qt_out,
                                  int32_t iters);

/* Developer / test hook: the encoder self-attention kernels alone (k_attn.hip) on n_clips clips of T frames of uniform random
 * q | k [R][2D] and V^T [D][R] at width D (head_dim 52).  variant 0 = block-streaming kernel, two 4-wave workgroups per
 * (clip, head); 1 = its 7-wave shape; 2 = two query tiles per wave; 100 + abl = the LDS-resident kernel (abl 0 = as shipped,
 * other values = ablations with garbage results).  Returns ms per launch over `iters` launches (0 = one untimed launch), < 0 on
 * error; out (nullable) [R][D], R = n_clips * round_up(T, 8), receives the output as bf16 bit patterns. */
MSH_EXPORT float msh_test_enc_attention(int32_t variant, int32_t n_clips, int32_t T, int32_t D, int32_t heads, int32_t iters,
                                        uint16_t* out);

/* Test hook (an engine created through THIS library): copy min(bytes, size) bytes of a named decode buffer of the last msh_decode call ("cache_k", "cache_v":
 * bf16 [layers][clips][heads][Smax][head_dim]; "resid": fp32 [clips][hidden]; "cross_k", "cross_v": K^T / V^T of the last
 * msh_encode, [layers][hidden * keys] at 2 bytes (bf16) or 1 byte (fp8) per key) to host memory; returns the buffer's
 * size in bytes, -1 on error.  "graph_captures" (dst unused) returns the number of decode-step hipGraphs this engine has
 * instantiated so far (captured steps are cached per batch shape).  No reference counterpart (ORT owns these tensors there). */
MSH_EXPORT int64_t msh_test_debug_read(msh_engine* e, const char* name, void* dst, uint64_t bytes);

#ifdef __cplusplus
}
#endif

#endif /* MOONSHINE_HIP_DEV_H */
