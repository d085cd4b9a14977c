// Development hooks (include/moonshine_hip_dev.h): kernel-alone test entry points and microbenchmarks over the internal
// launchers.  Linked into lib/libmoonshine_dev.so ONLY -- the product library exports none of these (moonshine_amd/build.py).
#include "../../include/moonshine_hip_dev.h"

#include <stdio.h>
#include <string.h>

#include <mutex>
#include <stdexcept>
#include <vector>

#include "engine.h"
#include "kernels.h"

float msh_test_gemm_microbench(int32_t M, int32_t N, int32_t K, int64_t lda, int32_t cfg, int32_t abl, int32_t iters) {
  try {
    return msh::gemm_microbench(M, N, K, lda, cfg, abl, iters);
  } catch (const std::exception& ex) {
    fprintf(stderr, "gemm_microbench: %s\n", ex.what());
    return -1.0f;
  }
}

float msh_test_mlp_microbench(int32_t R, int32_t D, int32_t F, int32_t iters, int32_t abl) {
  try {
    return msh::mlp_microbench(R, D, F, iters, abl);
  } catch (const std::exception& ex) {
    fprintf(stderr, "mlp_microbench: %s\n", ex.what());
    return -1.0f;
  }
}

float msh_test_qkv_panel(int32_t R, int32_t D, int32_t iters, uint16_t* out_qk, uint16_t* out_vt, float* out_h, float* out_w,
                         int32_t* out_pos) {
  try {
    return msh::qkv_panel_microbench(R, D, iters, out_qk, out_vt, out_h, out_w, out_pos);
  } catch (const std::exception& ex) {
    fprintf(stderr, "qkv_panel: %s\n", ex.what());
    return -1.0f;
  }
}

int32_t msh_test_mlp_run(float* h, int32_t R, int32_t D, int32_t F, const float* w1, const float* gamma, const float* b1,
                         const float* w2, const float* b2) {
  try {
    msh::mlp_fused_host(h, R, D, F, w1, gamma, b1, w2, b2);
    return MSH_OK;
  } catch (const std::exception& ex) {
    fprintf(stderr, "mlp_run: %s\n", ex.what());
    return MSH_ERR_UNKNOWN;
  }
}

int32_t msh_test_mlp_oproj_run(float* h, int32_t R, int32_t D, int32_t F, const float* w1, const float* gamma, const float* b1,
                               const float* w2, const float* b2, const float* ao, const float* wo) {
  try {
    msh::mlp_fused_host(h, R, D, F, w1, gamma, b1, w2, b2, ao, wo);
    return MSH_OK;
  } catch (const std::exception& ex) {
    fprintf(stderr, "mlp_oproj_run: %s\n", ex.what());
    return MSH_ERR_UNKNOWN;
  }
}

extern "C" int64_t msh_internal_debug_read(msh_engine* e, const char* name, void* dst, uint64_t bytes);   // msh_api.cpp, not exported

int64_t msh_test_debug_read(msh_engine* e, const char* name, void* dst, uint64_t bytes) { return msh_internal_debug_read(e, name, dst, bytes); }

float msh_test_enc_attention(int32_t variant, int32_t n_clips, int32_t T, int32_t D, int32_t heads, int32_t iters, uint16_t* out) {
  try {
    return msh::enc_attention_microbench(variant, n_clips, T, D, heads, iters, out);
  } catch (const std::exception& ex) {
    fprintf(stderr, "enc_attention: %s\n", ex.what());
    return -1.0f;
  }
}

int32_t msh_test_mlp_oproj_y_run(float* h, int32_t R, int32_t D, int32_t F, const float* w1, const float* gamma, const float* b1,
                                 const float* w2, const float* b2, const float* ao, const float* wo, uint16_t* y_fm) {
  try {
    msh::mlp_fused_host(h, R, D, F, w1, gamma, b1, w2, b2, ao, wo, y_fm);
    return MSH_OK;
  } catch (const std::exception& ex) {
    fprintf(stderr, "mlp_oproj_y_run: %s\n", ex.what());
    return MSH_ERR_UNKNOWN;
  }
}

float msh_test_crossq2(const float* x, const float* wq, const float* wk, int32_t M, int32_t D, float* qt_out, int32_t iters) {
  try {
    return msh::crossq2_host(x, wq, wk, M, D, qt_out, iters);
  } catch (const std::exception& ex) {
    fprintf(stderr, "crossq2: %s\n", ex.what());
    return -1.0f;
  }
}

float msh_test_cross_absorbed(const float* qt, const float* enc, int64_t R, const int32_t* Ts, const int32_t* row_starts,
                              int32_t M, int32_t D, float* ctx_out, int32_t iters) {
  try {
    return msh::cross_absorbed_host(qt, enc, (long)R, Ts, row_starts, M, D, ctx_out, iters);
  } catch (const std::exception& ex) {
    fprintf(stderr, "cross_absorbed: %s\n", ex.what());
    return -1.0f;
  }
}

// Self-test of the device allocator (meant for MSH_GUARD_ALLOC=1): odd-sized buffers, pageable H2D / D2H through the
// utility stream, zero-fill, D2D.  Returns 0 when every byte came back, a negative step number otherwise.
int32_t msh_test_device_alloc(void) {
  try {
    const size_t sizes[] = {1000, 4096, 5000, 1 << 20, (3 << 20) + 48};
    int step = 0;
    for (size_t n : sizes) {
      ++step;
      void *a = nullptr, *b = nullptr;
      {
        std::lock_guard<std::mutex> lock(msh::device_structure_mutex());
        a = msh::device_alloc(n);
        b = msh::device_alloc(n);
      }
      std::vector<unsigned char> h(n), back(n, 0);
      for (size_t i = 0; i < n; ++i) h[i] = (unsigned char)(i * 131 + 7);
      msh::zero_blocking(a, n);
      msh::copy_blocking(back.data(), a, n, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < n; ++i)
        if (back[i] != 0) return -(step * 10 + 1);
      msh::copy_blocking(a, h.data(), n, hipMemcpyHostToDevice);
      msh::copy_blocking(b, a, n, hipMemcpyDeviceToDevice);
      msh::copy_blocking(back.data(), b, n, hipMemcpyDeviceToHost);
      if (back != h) return -(step * 10 + 2);
      // an interior slice, as the engines' staging copies do
      if (n > 300) {
        msh::copy_blocking(static_cast<char*>(a) + 128, h.data(), 100, hipMemcpyHostToDevice);
        msh::copy_blocking(back.data(), a, n, hipMemcpyDeviceToHost);
        if (memcmp(back.data() + 128, h.data(), 100) != 0 || back[127] != h[127] || back[228] != h[228]) return -(step * 10 + 3);
      }
      std::lock_guard<std::mutex> lock(msh::device_structure_mutex());
      msh::device_free(a);
      msh::device_free(b);
    }
    fprintf(stderr, "msh_test_device_alloc: ok (guard allocator %s)\n", msh::guard_alloc_enabled() ? "on" : "off");
    return 0;
  } catch (const std::exception& ex) {
    fprintf(stderr, "msh_test_device_alloc: %s\n", ex.what());
    return -1000;
  }
}
