// Shared declarations for the MI355X Moonshine engine (device side).
// gfx950 only: 64-wide wavefronts, bf16 MFMA 16x16x32, fp32 accumulate.
#pragma once

#include <stdlib.h>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace msh {

typedef uint16_t bf16_t;  // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define MSH_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      throw ::msh::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + \
                            __FILE__ + ":" + std::to_string(__LINE__) + ")");               \
    }                                                                                        \
  } while (0)

// Every kernel launch goes through this: a launch the runtime refuses (grid above the limits, too much LDS for the
// kernel, a template instance that was never compiled) is reported HERE, by name and line, instead of surfacing later as a
// sticky error at an unrelated synchronise -- or as garbage ids.  hipGetLastError is legal during stream capture and
// costs a thread-local read; launches in the steady state are graph replays, which do not pass through here at all.
// MSH_TRACE_LAUNCH=1 (diagnostic): every eager launch prints its kernel to stderr and waits for it, so that a GPU memory
// fault is attributed to the kernel that caused it (launches inside a stream capture are only printed).
// MSH_TRACE_LAUNCH=2 also prints grid, block and every pointer / integer argument BEFORE the launch (with
// MSH_GUARD_ALLOC=2, which logs every allocation, an out-of-bounds access can be worked out from the log alone).
void trace_launch(const char* kernel, const char* file, int line, hipStream_t s);
int trace_launch_level();
void trace_launch_begin(const char* kernel, dim3 grid, dim3 block);
template <class T>
inline void trace_launch_arg(const T& v) {
  if constexpr (std::is_pointer<T>::value) fprintf(stderr, " %p", (const void*)v);
  else if constexpr (std::is_integral<T>::value) fprintf(stderr, " %lld", (long long)v);
  else if constexpr (std::is_floating_point<T>::value) fprintf(stderr, " %g", (double)v);
  else {   // a struct of arguments (epilogue functors): its 8-byte words
    fprintf(stderr, " {");
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
    for (size_t i = 0; i + 8 <= sizeof(T) && i < 96; i += 8) {
      unsigned long long w;
      memcpy(&w, b + i, 8);
      fprintf(stderr, "%s0x%llx", i ? " " : "", w);
    }
    fprintf(stderr, "}");
  }
}
template <class... A>
inline void trace_launch_args(const char* kernel, dim3 grid, dim3 block, const A&... a) {
  if (trace_launch_level() < 2) return;
  trace_launch_begin(kernel, grid, block);
  (trace_launch_arg(a), ...);
  fprintf(stderr, "\n");
  fflush(stderr);
}
#define MSH_LAUNCH(kernel, grid, block, lds, stream, ...)                     \
  do {                                                                        \
    ::msh::trace_launch_args(#kernel, grid, block, __VA_ARGS__);              \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);        \
    MSH_HIP(hipGetLastError());                                               \
    ::msh::trace_launch(#kernel, __FILE__, __LINE__, stream);                 \
  } while (0)

// Developer switches (kernel variants, ablations, A/B thresholds) are read through dev_getenv, which answers
// only when MSH_DEV_KNOBS=1 is set as well: a production process cannot pick up an unmeasured path from a stray MSH_* variable.
// The tests and the tools set it.  Diagnostics (MSH_GUARD_*, MSH_TRACE_LAUNCH, MSH_*_TIMING) are read with plain getenv.
inline const char* dev_getenv(const char* name) {
  static const bool on = [] {
    const char* e = getenv("MSH_DEV_KNOBS");
    return e != nullptr && e[0] == '1';
  }();
  return on ? getenv(name) : nullptr;
}

// Blocking copies / zero-fills that stay OFF the legacy (null) stream: hipMemcpy / hipMemset / hipDeviceSynchronize
// touch it, and the legacy stream may not be used while ANOTHER host thread captures a decode-step graph on its own
// stream ("operation would make the legacy stream depend on a capturing blocking stream") -- which is exactly what
// several engines or lanes in one process do.  They run on ONE long-lived non-blocking utility stream per device
// (serialised by a mutex) and wait for it, so the result is visible to every later launch on any stream.  One stream,
// not one per call: HIP hands streams their hardware queue round-robin out of a small pool (GPU_MAX_HW_QUEUES, 4 by
// default), and a churn of short-lived streams made two engine lanes land on the same queue, i.e. run one after the
// other (profiles/r03e_kernel_trace: lanes on queues 3, 4, 4).
// Device allocation / release and graph capture never run at the same time in one process: hipMalloc / hipFree
// synchronise the whole device behind the scenes, and doing that from one host thread while another is between
// hipStreamBeginCapture and hipStreamEndCapture is where a (rare) crash of the multi-lane tests pointed.  Both sides
// take this mutex (one per device since round 5); it is only ever contended while engines of ONE device warm up.
std::mutex& device_structure_mutex(int device = -1);   // -1: the calling thread's current device
// Every device allocation of the library (call with device_structure_mutex held).  Normally hipMalloc / hipFree.  With
// MSH_GUARD_ALLOC=1 (diagnostic) each buffer gets its own virtual-memory mapping whose END is the end of the mapped range,
// followed by an unmapped granule: a kernel that reads or writes 16 bytes or more past the end of ANY buffer takes a GPU
// memory fault there and then (tools/gpu_debug_fault.sh names the kernel) instead of silently reading a neighbour.
void* device_alloc(size_t bytes);
void device_free(void* p);
bool guard_alloc_enabled();
void copy_blocking(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
void zero_blocking(void* p, size_t bytes);

// ---- bf16 <-> fp32 (round-to-nearest-even), usable on host and device ----
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  union {
    float f;
    uint32_t u;
  } v;
  v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);  // quiet NaN
  uint32_t lsb = (v.u >> 16) & 1u;
  v.u += 0x7fffu + lsb;
  return (bf16_t)(v.u >> 16);
}
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
  union {
    float f;
    uint32_t u;
  } v;
  v.u = ((uint32_t)h) << 16;
  return v.f;
}

// Device-side packing uses the gfx950 hardware conversion (v_cvt_pk_bf16_f32, round-to-nearest-even): one
// instruction instead of the ~10 (with an exec-masked NaN branch) the portable bit-twiddling compiles to.
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_native f = {lo, hi};
  const bf16x2_native b = __builtin_convertvector(f, bf16x2_native);
  return *reinterpret_cast<const uint32_t*>(&b);
}

// ---- per-batch clip geometry, shared by host planner and kernels ----
// Rows of the packed encoder stream: clip b owns rows [row_start, row_start + rows)
// of every [*, D] activation; conv2 output rows are 2x, conv1 output rows 6x that
// (so the strided conv-as-GEMM views have one uniform row stride over the batch).
struct ClipMeta {
  int32_t row_start;  // first row in the [R, D] stream
  int32_t rows;       // R_b (multiple of 8): padded frame count
  int32_t T;          // valid encoder frames (conv3 length)
  int32_t L1;         // valid conv1 frames
  int32_t L2;         // valid conv2 frames
  int32_t n_samples;  // audio samples fed to the model
  int32_t kv_start;   // offset (in keys) of this clip in the cross K^T/V^T buffers
  int32_t Tk;         // padded key count (multiple of 8) of the cross K^T/V^T rows
  int32_t max_len;    // decode step budget: ceil(n/16000 * max_tokens_per_second)
  int32_t pad_;
};

}  // namespace msh
