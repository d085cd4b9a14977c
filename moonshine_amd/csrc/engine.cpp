#include "engine.h"

#include <atomic>

#include "host_utils.h"

#include <mutex>
#include <unordered_map>

#include <math.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>

namespace msh {

namespace {
struct UtilStreams {
  std::mutex mu;
  std::map<int, hipStream_t> by_device;
  hipStream_t get() {  // call with mu held
    int dev = 0;
    MSH_HIP(hipGetDevice(&dev));
    auto it = by_device.find(dev);
    if (it != by_device.end()) return it->second;
    hipStream_t s = nullptr;
    MSH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    by_device[dev] = s;
    return s;
  }
};
UtilStreams& util_streams() {
  static UtilStreams* u = new UtilStreams();  // never destroyed: the HIP runtime may already be gone at exit
  return *u;
}
}  // namespace

// developer knob MSH_XATTN_QT=1 (at load AND at run time): the keys-side queries of the absorbed cross-attention from the
// merged weight Wqk (one wide decode GEMM) instead of the two-stage kernel (k_crossq.hip)
int xattn_qt_mode() {   // 0 = two-stage kernel, 1 = merged weight, 2 = two-stage kernel with the merged weight uploaded as well (probe)
  static const int v = [] {
    const char* e = dev_getenv("MSH_XATTN_QT");
    return e != nullptr ? atoi(e) : 0;
  }();
  return v;
}
bool xattn_merged_qt() { return xattn_qt_mode() == 1; }
// the absorbed cross-attention's encoder rows with the non-temporal policy: by default when the engine is one of several lanes
// on the GPU; MSH_XATTN_NT=0 / 1 forces it off / on (developer knob)
bool xattn_stream_nt(bool shared_gpu) {
  static const int forced = [] {
    const char* e = dev_getenv("MSH_XATTN_NT");
    return e != nullptr ? (e[0] == '0' ? 0 : 1) : -1;
  }();
  return forced >= 0 ? forced != 0 : shared_gpu;
}
// the encoder block kernels' outputs (fused MLP, QKV panel) with non-temporal stores: measured SLOWER, in the kernels themselves
// (0.62 / 0.33 against 0.57 / 0.30 ms) and in the overlapped bench (89.5 against 90.5 k audio-s/s; profiles/r5v_*), so it is off;
// MSH_ENC_STORE_NT=1 forces it on (probe)
constexpr bool kEncStoreNtInLanes = false;
bool enc_store_nt(bool shared_gpu) {
  static const int forced = [] {
    const char* e = dev_getenv("MSH_ENC_STORE_NT");
    return e != nullptr ? (e[0] == '0' ? 0 : 1) : -1;
  }();
  return forced >= 0 ? forced != 0 : (shared_gpu && kEncStoreNtInLanes);
}
// Developer knob MSH_XATTN_MIN_BATCH=n (> 0): mode 0 switches to the absorbed form per batch from n clips on, as round 4 did
// (A/B measurements only: a production engine has ONE form, Engine::set_cross_mode).  Where the absorbed form pays: one
// workgroup per clip, so below ~200 clips most of the chip idles through the kernel and its two wider GEMMs cost more than
// the halved stream saves (64 / 128 / 256 clips x 10 s: 8.7 / 12.0 / 17.8 us against 9 / 15 / 28.9) -- the host layer's
// `auto` picks it when the configured sub-batch size is >= 192 (transcriber.cpp).
int xattn_min_batch() {
  static const int v = [] {
    const char* e = dev_getenv("MSH_XATTN_MIN_BATCH");
    return e != nullptr ? atoi(e) : 0;
  }();
  return v;
}

// decode steps per graph replay in the steady state of the decode loop (MSH_DEC_GRAPH_STEPS; 1 = one replay per step)
int graph_steps() {
  static const int v = [] {
    const char* e = dev_getenv("MSH_DEC_GRAPH_STEPS");
    const int n = e != nullptr ? atoi(e) : 8;
    return n == 1 || n == 2 || n == 4 || n == 8 ? n : 8;
  }();
  return v;
}

int trace_launch_level() {
  static const int level = [] {
    const char* e = getenv("MSH_TRACE_LAUNCH");
    return e != nullptr ? atoi(e) : 0;
  }();
  return level;
}
void trace_launch_begin(const char* kernel, dim3 grid, dim3 block) {
  fprintf(stderr, "[msh args] %.60s <<<(%u,%u,%u),(%u)>>>", kernel, grid.x, grid.y, grid.z, block.x);
}
void trace_launch(const char* kernel, const char* file, int line, hipStream_t s) {
  if (trace_launch_level() < 1) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cs);
  const char* base = strrchr(file, '/');
  fprintf(stderr, "[msh launch] %.100s (%s:%d)%s\n", kernel, base ? base + 1 : file, line,
          cs == hipStreamCaptureStatusNone ? "" : " [captured]");
  fflush(stderr);
  if (cs == hipStreamCaptureStatusNone) MSH_HIP(hipStreamSynchronize(s));
}

// ---- device allocation (msh_common.h) ----
namespace {
struct GuardBlock {
  hipMemGenericAllocationHandle_t handle;
  void* va;
  size_t va_size, map_size;
};
std::unordered_map<void*, GuardBlock>& guard_blocks() {
  static auto* m = new std::unordered_map<void*, GuardBlock>();
  return *m;
}
}  // namespace
bool guard_alloc_enabled() {
  static const bool on = [] {
    const char* e = getenv("MSH_GUARD_ALLOC");
    return e != nullptr && (e[0] == '1' || e[0] == '2');   // 2: also log every allocation
  }();
  return on;
}
void* device_alloc(size_t bytes) {
  if (bytes == 0) bytes = 16;
  void* p = nullptr;
  if (!guard_alloc_enabled()) {
    MSH_HIP(hipMalloc(&p, bytes));
    return p;
  }
  int dev = 0;
  MSH_HIP(hipGetDevice(&dev));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  MSH_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (gran == 0) gran = (size_t)2 << 20;
  GuardBlock b{};
  b.map_size = (bytes + gran - 1) / gran * gran;
  b.va_size = b.map_size + gran;   // the granule after the mapping stays unmapped
  MSH_HIP(hipMemAddressReserve(&b.va, b.va_size, gran, nullptr, 0));
  MSH_HIP(hipMemCreate(&b.handle, b.map_size, &prop, 0));
  MSH_HIP(hipMemMap(b.va, b.map_size, 0, b.handle, 0));
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  MSH_HIP(hipMemSetAccess(b.va, b.map_size, &acc, 1));
  // the buffer keeps hipMalloc's 256-byte alignment by default (MSH_GUARD_ALIGN=16 tightens the net to 16 bytes, but then
  // the bases are less aligned than anything the product ever sees)
  static const size_t align = [] {
    const char* e = getenv("MSH_GUARD_ALIGN");
    const long v = e ? atol(e) : 256;
    return (size_t)(v >= 16 && (v & (v - 1)) == 0 ? v : 256);
  }();
  const size_t tail = (bytes + align - 1) & ~(align - 1);
  // the bytes in front of the buffer (same mapping) read as zeros, like fresh hipMalloc memory -- or, with MSH_GUARD_POISON=1,
  // as 0xFF bytes (bf16 / fp32 NaN, int -1): a kernel whose RESULT depends on what lies before its buffer then shows it
  static const int fill = [] {
    const char* e = getenv("MSH_GUARD_POISON");
    return e != nullptr && e[0] == '1' ? 0xFF : 0;
  }();
  MSH_HIP(hipMemset(b.va, fill, b.map_size));
  MSH_HIP(hipDeviceSynchronize());
  p = static_cast<char*>(b.va) + (b.map_size - tail);
  guard_blocks()[p] = b;
  static const bool log = [] {
    const char* e = getenv("MSH_GUARD_ALLOC");
    return e != nullptr && e[0] == '1' && e[1] == '\0' ? false : true;
  }();
  if (log) fprintf(stderr, "[msh alloc] %p .. %p (%zu bytes)\n", p, (void*)(static_cast<char*>(p) + bytes), bytes);
  return p;
}
void device_free(void* p) {
  if (p == nullptr) return;
  if (!guard_alloc_enabled()) {
    (void)hipFree(p);
    return;
  }
  auto it = guard_blocks().find(p);
  if (it == guard_blocks().end()) return;
  const GuardBlock b = it->second;
  guard_blocks().erase(it);
  (void)hipDeviceSynchronize();
  (void)hipMemUnmap(b.va, b.map_size);
  (void)hipMemRelease(b.handle);
  // The address range is NOT handed back: on this stack a range that is reserved and mapped again straight away can still
  // be served from stale translations / cache lines of its previous mapping (seen as index arrays full of the old
  // buffer's activations).  A diagnostic run leaks address space instead -- and a use-after-free keeps pointing at
  // unmapped addresses for the rest of the process.
}

std::mutex& device_structure_mutex(int device) {
  // one per DEVICE: hipMalloc / hipFree synchronise their own device only, and a capture is a property of a stream of one
  // device -- eight engines on eight GPUs of one process (options num_gpus / devices) warm up side by side, not in a queue
  static std::mutex* m = new std::mutex[65];
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 64;
  return m[device >= 0 && device < 64 ? device : 64];
}

// One large host -> device upload per device at a time (Engine::encode): four lanes that start their sub-batches together
// otherwise share the link, all four uploads end together (4 x 164 MB: ~26 ms) and no lane computes before that; in turn, the
// first lane starts after a quarter of it and the others upload beside its kernels.
std::mutex& device_upload_mutex(int device) {
  static std::mutex* m = new std::mutex[65];
  return m[device >= 0 && device < 64 ? device : 64];
}

void copy_blocking(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (bytes == 0) return;
  UtilStreams& u = util_streams();
  std::lock_guard<std::mutex> lock(u.mu);
  hipStream_t s = u.get();
  MSH_HIP(hipMemcpyAsync(dst, src, bytes, kind, s));
  MSH_HIP(hipStreamSynchronize(s));
}
void zero_blocking(void* p, size_t bytes) {
  if (bytes == 0) return;
  UtilStreams& u = util_streams();
  std::lock_guard<std::mutex> lock(u.mu);
  hipStream_t s = u.get();
  MSH_HIP(hipMemsetAsync(p, 0, bytes, s));
  MSH_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------------------------
// Developer probe MSH_BUF_SKEW_KB=n: the k-th workspace allocation of the process starts (k mod 8) * n KiB into its
// device allocation.  Large hipMalloc results share their alignment, so streams that walk two buffers at the same pace (a
// kernel's input rows and its output rows) keep the same low address bits for the whole launch; where that puts both on
// the same memory channel depends on the physical pages behind them.  (The probe answers whether the box-to-box spread of
// the encoder panel kernels, DESIGN.md 3b / 3c, is that.)
static size_t buf_skew_bytes() {
  static const size_t kb = [] {
    const char* e = dev_getenv("MSH_BUF_SKEW_KB");
    const long v = e != nullptr ? atol(e) : 0;
    return (size_t)(v > 0 && !guard_alloc_enabled() ? v : 0);
  }();
  if (kb == 0) return 0;
  static std::atomic<unsigned> counter{0};
  return (size_t)(counter.fetch_add(1) % 8u) * kb * 1024;
}
bool DevBuf::reserve(size_t bytes) {
  if (bytes <= cap && p != nullptr) return false;
  // a little slack so ragged batches do not thrash (none under the guard allocator: it would hide over-reads)
  size_t want = guard_alloc_enabled() ? bytes : bytes + bytes / 8 + 256;
  const size_t skew = buf_skew_bytes();
  std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
  void* nr = device_alloc(want + skew);
  void* np = static_cast<char*>(nr) + skew;
  // The zero-fill must be complete before the first kernel or copy on an engine stream writes the new buffer (a plain
  // hipMemset is asynchronous null-stream work those streams do not wait for: seen as rare garbage logits).
  zero_blocking(np, want);
  if (raw) device_free(raw);
  raw = nr;
  p = np;
  cap = want;
  return true;
}
void DevBuf::release() {
  std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
  if (raw) device_free(raw);
  raw = nullptr;
  p = nullptr;
  cap = 0;
}

// ------------------------------------------------------------------------------------------------
Engine::Engine(DryRun) : device_(-1), dry_run_(true) {}

ModelConfig Engine::check_weights(const SafeTensors& st, int expect_arch) {
  Engine e(DryRun{});
  e.load_weights(st, expect_arch);
  return e.cfg_;
}

Engine::Engine(int device) : device_(device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    throw HipError("no HIP device available: the MI355X engine has no CPU fallback (" +
                   std::string(hipGetErrorString(e)) + ")");
  if (device < 0 || device >= n) throw HipError("invalid device index " + std::to_string(device));
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  {
    // first command now, not at the first encode(): HIP binds a stream to its hardware queue lazily and round-robin, so
    // engines (lanes) created one after the other get consecutive queues only if each touches its stream right away
    {
      std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
      stream_probe_ = device_alloc(256);
    }
    MSH_HIP(hipMemsetAsync(stream_probe_, 0, 256, stream_));
    MSH_HIP(hipStreamSynchronize(stream_));
  }
  if (const char* sk = dev_getenv("MSH_ALLOC_SKEW_KB")) {   // developer probe: shift every later allocation of this engine (placement sensitivity)
    const long kb = atol(sk);
    if (kb > 0) {
      std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
      weight_allocs_.push_back(device_alloc((size_t)kb << 10));
    }
  }
  const char* ng = dev_getenv("MSH_NO_GRAPH");
  if (ng != nullptr && ng[0] == '1') use_graph_ = false;
  const char* dg = dev_getenv("MSH_DEC_GROUPS");
  if (dg != nullptr) dec_groups_ = atoi(dg);
}

Engine::~Engine() {
  if (dry_run_) return;   // nothing was ever allocated
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  groups_.clear();
  if (enc_done_) (void)hipEventDestroy(enc_done_);
  if (enc_fork_) (void)hipEventDestroy(enc_fork_);
  if (enc_join_) (void)hipEventDestroy(enc_join_);
  if (enc_stream2_) (void)hipStreamDestroy(enc_stream2_);
  {
    std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
    if (stream_probe_) device_free(stream_probe_);
    for (void* p : weight_allocs_) device_free(p);
    weight_allocs_.clear();
  }
  for (hipEvent_t ev : event_pool_) (void)hipEventDestroy(ev);
  for (auto& r : prof_pending_) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  DevBuf* bufs[] = {&clips_d_, &clip_ptrs_d_, &pcm_stage_, &audio_bf16_, &row_pos_, &row_clip_, &x1_, &x2_, &H_,
                    &Y_, &QKV_, &VTe_, &AO_, &Z_, &ENC_, &ENC32_, &gn_part_, &gn_rows_, &gn_stats_, &gn_table_, &KT_, &VT_, &cross_probs_};
  for (DevBuf* b : bufs) b->release();
  if (pcm_pinned_) (void)hipHostFree(pcm_pinned_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::synchronize() {
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamSynchronize(stream_));
}

// Weights come out of a few large slabs instead of one hipMalloc each: ~150 tensors of 1.6 KB .. 2.8 MB would otherwise be ~150
// separate mappings of odd sizes, and where they land decides how large the page-table fragments behind them are.  (The
// decode step walks ~100 MB of weights next to ~300 MB of caches; round 3 and round 4 both saw the SAME binaries run 15-25 %
// slower in some processes than in others)  256-byte aligned like hipMalloc's results.  MSH_WEIGHT_ARENA=0
// or the guard allocator (every buffer its own mapping, that is its point): one allocation per tensor as before.
void* Engine::weight_alloc(size_t bytes) {
  static const bool arena = [] {
    const char* e = dev_getenv("MSH_WEIGHT_ARENA");
    return !(e != nullptr && e[0] == '0') && !guard_alloc_enabled();
  }();
  constexpr size_t kSlab = (size_t)64 << 20;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
  if (!arena || bytes > kSlab / 2) {
    void* p = device_alloc(bytes);
    weight_allocs_.push_back(p);
    return p;
  }
  if (slab_ == nullptr || slab_used_ + bytes > kSlab) {
    slab_ = static_cast<char*>(device_alloc(kSlab));
    weight_allocs_.push_back(slab_);
    slab_used_ = 0;
  }
  void* p = slab_ + slab_used_;
  slab_used_ += bytes;
  return p;
}

void Engine::upload(const std::vector<float>& src, float** dst) {
  if (dry_run_) return;
  void* p = weight_alloc(src.size() * sizeof(float));
  copy_blocking(p, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
  *dst = reinterpret_cast<float*>(p);
}

void Engine::upload_bf16(const std::vector<float>& src, bf16_t** dst) {
  if (dry_run_) return;
  std::vector<bf16_t> tmp(src.size());
  for (size_t i = 0; i < src.size(); ++i) tmp[i] = f32_to_bf16(src[i]);
  void* p = weight_alloc(tmp.size() * sizeof(bf16_t));
  copy_blocking(p, tmp.data(), tmp.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
  *dst = reinterpret_cast<bf16_t*>(p);
}

// bf16 upload of a [rows][K] matrix in the MFMA-fragment-major order of the decode GEMMs (kernels.h fm16)
void Engine::upload_bf16_fm(const std::vector<float>& src, int rows, int K, bf16_t** dst) {
  if ((rows & 15) != 0 || (K & 31) != 0 || src.size() != (size_t)rows * K)
    throw std::runtime_error("decode weight [" + std::to_string(rows) + ", " + std::to_string(K) +
                             "] cannot be packed for the MFMA decode kernels (rows % 16, K % 32)");
  if (dry_run_) return;
  std::vector<bf16_t> tmp(src.size());
  const int ks = K >> 5;
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) tmp[(size_t)fm16(r, k, ks)] = f32_to_bf16(src[(size_t)r * K + k]);
  void* p = weight_alloc(tmp.size() * sizeof(bf16_t));
  copy_blocking(p, tmp.data(), tmp.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
  *dst = reinterpret_cast<bf16_t*>(p);
}

// ------------------------------------------------------------------------------------------------
// Weights.  Tensor names / shapes: HuggingFace MoonshineForConditionalGeneration state_dict
// (transformers modeling_moonshine.py:520-540 stem, :265-274 attention, :74-75 / :89-90 MLPs,
// :836-850 tied head).  Layout changes made here, once, at load:
//   conv1  [D,1,127]   -> [D][128] (tap 127 = 0)                  GEMM over the raw sample stream, lda = 64
//   conv2  [2D,D,7]    -> [2D][7][D]  (tap-major, channel-minor)  matches the channels-last window
//   conv3  [D,2D,3]    -> [D][3][2D]
//          both then re-ordered along K to [C/32][taps][32]: the order in which the conv GEMMs walk their k-slices
//          (conv_k_offset, gemm_common.h: the two uses of an input byte three slices apart instead of 13 .. 39)
//   q,k,v  3 x [D,D]   -> [3D][D] fused
//   cross k,v of all decoder layers -> [L*2D][D] (one GEMM per batch)
//   decoder fc1 [2F,D] -> rows interleaved (value_j, gate_j) so SwiGLU pairs sit in one lane
void Engine::share_weights_from(const Engine& o) {
  if (loaded_) throw std::runtime_error("weights already loaded");
  if (!o.loaded_ || o.device_ != device_) throw std::runtime_error("share_weights_from: owner not loaded / other device");
  cfg_ = o.cfg_;
  conv1_w_ = o.conv1_w_, conv2_w_ = o.conv2_w_, conv3_w_ = o.conv3_w_;
  conv_kperm_ = o.conv_kperm_;
  conv2_s1_ = o.conv2_s1_, conv2_b2_ = o.conv2_b2_, conv3_b_ = o.conv3_b_, enc_ln_ = o.enc_ln_;
  enc_ = o.enc_;
  dec_ = o.dec_;
  embed_bf16_ = o.embed_bf16_, embed_head_folded_ = o.embed_head_folded_, cross_kv_w_ = o.cross_kv_w_;
  cross_kv_panel_w_ = o.cross_kv_panel_w_;
  embed_f32_ = o.embed_f32_, dec_ln_ = o.dec_ln_;
  kv_qscale_ = o.kv_qscale_, kv_dq_ = o.kv_dq_;
  kv_fp8_ = o.kv_fp8_;
  cross_mode_ = o.cross_mode_;
  uniform_kernels_ = o.uniform_kernels_;
  rope_cos_ = o.rope_cos_, rope_sin_ = o.rope_sin_;
  rope_max_pos_ = o.rope_max_pos_;
  loaded_ = true;  // weight_allocs_ stays empty: the owner frees
}

// [N][taps][C] (the window's order in memory) -> [N][C/32][taps][32] (the k-slice order of conv_k_offset)
static std::vector<float> conv_tap_inner(const std::vector<float>& w, int N, int taps, int C) {
  std::vector<float> r(w.size());
  const int cbs = C / 32;
  for (int n = 0; n < N; ++n)
    for (int cb = 0; cb < cbs; ++cb)
      for (int t = 0; t < taps; ++t)
        for (int e = 0; e < 32; ++e)
          r[(((size_t)n * cbs + cb) * taps + t) * 32 + e] = w[((size_t)n * taps + t) * C + cb * 32 + e];
  return r;
}

// ------------------------------------------------------------------------------------------------
void Engine::load_weights(const SafeTensors& st, int expect_arch) {
  if (!dry_run_) MSH_HIP(hipSetDevice(device_));
  if (loaded_) throw std::runtime_error("weights already loaded");
  st.used.clear();
  ModelConfig c;
  const StTensor& c1 = st.get("model.encoder.conv1.weight");
  if (c1.shape.size() != 3 || c1.shape[1] != 1 || c1.shape[2] != 127)
    throw std::runtime_error("conv1.weight has unexpected shape");
  c.hidden = (int)c1.shape[0];
  c.ffn = (int)st.get("model.encoder.layers.0.mlp.fc1.weight").shape[0];
  c.vocab = (int)st.get("model.decoder.embed_tokens.weight").shape[0];
  while (st.has("model.encoder.layers." + std::to_string(c.enc_layers) + ".mlp.fc1.weight")) ++c.enc_layers;
  while (st.has("model.decoder.layers." + std::to_string(c.dec_layers) + ".mlp.fc1.weight")) ++c.dec_layers;
  auto md = st.metadata;
  c.arch = md.count("arch") ? md["arch"] : (c.hidden == 288 ? "tiny" : c.hidden == 416 ? "base" : "custom");
  c.heads = md.count("heads") ? std::stoi(md["heads"]) : (c.arch == "micro" ? 4 : 8);
  if (expect_arch == 0 && c.hidden != 288)
    throw std::runtime_error("model_arch TINY requested but weights have hidden size " + std::to_string(c.hidden));
  if (expect_arch == 1 && c.hidden != 416)
    throw std::runtime_error("model_arch BASE requested but weights have hidden size " + std::to_string(c.hidden));
  const int D = c.hidden, F = c.ffn, V = c.vocab, dh = D / c.heads;
  if (D % 32 != 0 || D % c.heads != 0 || dh % 4 != 0 || dh > 64 || (dh != 52 && dh != 36 && dh != 16))
    throw std::runtime_error("unsupported model dimensions (hidden " + std::to_string(D) + ", heads " +
                             std::to_string(c.heads) + ")");
  if (F % 32 != 0 || V % 4 != 0) throw std::runtime_error("unsupported ffn / vocab size");
  cfg_ = c;

  auto expect_shape = [&](const std::string& name, std::vector<int64_t> shape) {
    if (st.get(name).shape != shape) throw std::runtime_error("unexpected shape for " + name);
  };
  // 1-D parameters (biases, norm scales): checked against the config like the matrices, so a file whose vectors are short
  // can never make a kernel read past an allocation
  auto vec = [&](const std::string& name, int64_t n) {
    expect_shape(name, {n});
    return st.to_f32(name);
  };

  {  // conv stem
    std::vector<float> w = st.to_f32("model.encoder.conv1.weight"), r((size_t)D * 128, 0.f);
    for (int n = 0; n < D; ++n)
      for (int k = 0; k < 127; ++k) r[(size_t)n * 128 + k] = w[(size_t)n * 127 + k];
    upload_bf16(r, &conv1_w_);
    expect_shape("model.encoder.conv2.weight", {2 * D, D, 7});
    w = st.to_f32("model.encoder.conv2.weight");
    {
      // GroupNorm(1 group) between conv1 and conv2 is folded into conv2 (EpiGnBiasGeluBf16): its scale gamma goes into
      // the weights per input channel, its shift beta and the per-clip mean into two per-output-channel vectors.
      // S1 sums the bf16-ROUNDED folded weights, i.e. exactly what the MFMA multiplies.
      const std::vector<float> gam = vec("model.encoder.groupnorm.weight", D), bet = vec("model.encoder.groupnorm.bias", D);
      const std::vector<float> b2 = vec("model.encoder.conv2.bias", 2 * D);
      r.assign((size_t)2 * D * 7 * D, 0.f);
      std::vector<float> s1((size_t)2 * D, 0.f), s2((size_t)2 * D, 0.f);
      for (int n = 0; n < 2 * D; ++n) {
        double a1 = 0.0, a2 = 0.0;
        for (int ch = 0; ch < D; ++ch)
          for (int k = 0; k < 7; ++k) {
            const float wv = w[((size_t)n * D + ch) * 7 + k];
            const float folded = wv * gam[ch];
            r[((size_t)n * 7 + k) * D + ch] = folded;
            a1 += (double)bf16_to_f32(f32_to_bf16(folded));
            a2 += (double)wv * bet[ch];
          }
        s1[n] = (float)a1;
        s2[n] = (float)(a2 + b2[n]);
      }
      conv_kperm_ = [] {   // MSH_CONV_KORDER=0: the tap-major k-order of rounds 1-5 (A/B switch)
        const char* e = dev_getenv("MSH_CONV_KORDER");
        return (e != nullptr && e[0] == '0') ? 0 : 1;
      }();
      if (D % 32 != 0) conv_kperm_ = 0;
      if (conv_kperm_) r = conv_tap_inner(r, 2 * D, 7, D);
      upload_bf16(r, &conv2_w_);
      upload(s1, &conv2_s1_);
      upload(s2, &conv2_b2_);
    }
    expect_shape("model.encoder.conv3.weight", {D, 2 * D, 3});
    w = st.to_f32("model.encoder.conv3.weight");
    r.assign((size_t)D * 3 * 2 * D, 0.f);
    for (int n = 0; n < D; ++n)
      for (int ch = 0; ch < 2 * D; ++ch)
        for (int k = 0; k < 3; ++k) r[((size_t)n * 3 + k) * 2 * D + ch] = w[((size_t)n * 2 * D + ch) * 3 + k];
    if (conv_kperm_) r = conv_tap_inner(r, D, 3, 2 * D);
    upload_bf16(r, &conv3_w_);
    upload(vec("model.encoder.conv3.bias", D), &conv3_b_);
    upload(vec("model.encoder.layer_norm.weight", D), &enc_ln_);
  }
  auto fuse = [&](std::initializer_list<std::string> names) {
    std::vector<float> out;
    for (const std::string& n : names) {
      expect_shape(n, {D, D});
      std::vector<float> w = st.to_f32(n);
      out.insert(out.end(), w.begin(), w.end());
    }
    return out;
  };
  enc_.resize(c.enc_layers);
  for (int l = 0; l < c.enc_layers; ++l) {
    const std::string p = "model.encoder.layers." + std::to_string(l) + ".";
    EncLayerW& L = enc_[l];
    {
      std::vector<float> qkv = fuse({p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"});
      // The attention scale rsqrt(dh) and the exp2 domain's log2(e) are folded into the QUERY projection (hf:171-193 multiplies
      // the scores by `scaling`; RoPE is a rotation, so scaling q before it is the same thing): the encoder attention kernels
      // exponentiate MFMA outputs directly, 16 multiply-adds per 16 x 64 score tile less in a VALU-bound kernel (k_attn.hip).
      // One bf16 rounding of (scale * w) instead of w: the same relative error.
      {
        const float qs = (1.0f / sqrtf((float)c.head_dim())) * 1.4426950408889634f;
        for (size_t i = 0; i < (size_t)D * D; ++i) qkv[i] *= qs;
      }
      upload_bf16(qkv, &L.wqkv);
      if (qkv_panel_supported(D, c.head_dim(), c.rot_pairs()) && !dry_run_) {
        const std::vector<float> gam = vec(p + "input_layernorm.weight", D);
        std::vector<bf16_t> packed(panel_packed_elems(3 * D, D));
        pack_panel_weights(qkv.data(), gam.data(), 3 * D, D, packed.data());
        void* dp = weight_alloc(packed.size() * sizeof(bf16_t));
        copy_blocking(dp, packed.data(), packed.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
        L.qkv_panel = reinterpret_cast<bf16_t*>(dp);
      }
    }
    upload_bf16(fuse({p + "self_attn.o_proj.weight"}), &L.wo);
    expect_shape(p + "mlp.fc1.weight", {F, D});
    expect_shape(p + "mlp.fc2.weight", {D, F});
    upload_bf16(st.to_f32(p + "mlp.fc1.weight"), &L.fc1);
    upload_bf16(st.to_f32(p + "mlp.fc2.weight"), &L.fc2);
    upload(vec(p + "mlp.fc1.bias", F), &L.b1);
    upload(vec(p + "mlp.fc2.bias", D), &L.b2);
    upload(vec(p + "input_layernorm.weight", D), &L.ln1);
    upload(vec(p + "post_attention_layernorm.weight", D), &L.ln2);
    if (mlp_fused_supported(D, F) && !dry_run_) {   // the same block once more, packed for the fused kernel (k_mlp.hip)
      const std::vector<float> w1 = st.to_f32(p + "mlp.fc1.weight"), w2 = st.to_f32(p + "mlp.fc2.weight");
      const std::vector<float> g = vec(p + "post_attention_layernorm.weight", D), b1 = vec(p + "mlp.fc1.bias", F);
      // (with the attention output projection in front: the kernel forms H + AO Wo^T in its accumulators first)
      const std::vector<float> wo = fuse({p + "self_attn.o_proj.weight"});
      std::vector<bf16_t> packed(mlp_packed_elems(D, F, true));
      pack_mlp_weights(w1.data(), g.data(), b1.data(), w2.data(), D, F, packed.data(), wo.data());
      void* dp = weight_alloc(packed.size() * sizeof(bf16_t));
      copy_blocking(dp, packed.data(), packed.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
      L.mlp = reinterpret_cast<bf16_t*>(dp);
    }
  }
  {
    expect_shape("model.decoder.embed_tokens.weight", {V, D});
    std::vector<float> e = st.to_f32("model.decoder.embed_tokens.weight");
    upload(e, &embed_f32_);
    upload_bf16(e, &embed_bf16_);  // tied LM head (configuration_moonshine.py:103)
    std::vector<float> g = vec("model.decoder.norm.weight", D);
    upload(g, &dec_ln_);
    // copy of the head with the final LayerNorm scale folded in, for the LN-fused small-batch head GEMM
    for (int v = 0; v < V; ++v)
      for (int d = 0; d < D; ++d) e[(size_t)v * D + d] *= g[d];
    upload_bf16_fm(e, V, D, &embed_head_folded_);
  }
  dec_.resize(c.dec_layers);
  std::vector<float> cross;
  for (int l = 0; l < c.dec_layers; ++l) {
    const std::string p = "model.decoder.layers." + std::to_string(l) + ".";
    DecLayerW& L = dec_[l];
    // the decode GEMMs fuse LayerNorm into their A operand and expect its scale inside the weights:
    // LN(x) * W^T = ((x - mu) * rstd) * (W * diag(gamma))^T
    auto fold = [&](std::vector<float> w, const std::string& ln_name, int rows) {
      const std::vector<float> gam = vec(ln_name, D);
      for (int r = 0; r < rows; ++r)
        for (int d = 0; d < D; ++d) w[(size_t)r * D + d] *= gam[d];
      return w;
    };
    // decode weights go to the device in MFMA-fragment-major order (kernels.h); the cross-q weight additionally in
    // row-major for the small-batch kernel that fuses the query projection into the attention
    upload_bf16_fm(fold(fuse({p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}),
                        p + "input_layernorm.weight", 3 * D),
                   3 * D, D, &L.wqkv);
    upload_bf16_fm(fuse({p + "self_attn.o_proj.weight"}), D, D, &L.wo);
    {
      const std::vector<float> wq = fold(fuse({p + "encoder_attn.q_proj.weight"}), p + "post_attention_layernorm.weight", D);
      upload_bf16_fm(wq, D, D, &L.wq_c);
      upload_bf16(wq, &L.wq_c_rm);
    }
    const std::vector<float> wo_c = fuse({p + "encoder_attn.o_proj.weight"});
    upload_bf16_fm(wo_c, D, D, &L.wo_c);
    std::vector<float> kv = fuse({p + "encoder_attn.k_proj.weight", p + "encoder_attn.v_proj.weight"});
    cross.insert(cross.end(), kv.begin(), kv.end());
    if (cross_absorbed_supported(D, c.heads) && !dry_run_) {
      // Absorbed cross-attention (k_xattn.hip): the key projection moves onto the query, the value projection onto the
      // output projection, so that the attention itself runs over the encoder output.  Products in fp32, rounded once.
      //   wqk[(h, d)][k] = scale * sum_j Wk[h j][d] * Wq'[h j][k]      (Wq' = Wq * diag(gamma), scale = log2(e) / sqrt(dh))
      //   wvo[n][(h, d)] = sum_j Wo[n][h j] * Wv[h j][d]
      const int Hn = c.heads, dh = D / Hn;
      const std::vector<float> wqf = fold(fuse({p + "encoder_attn.q_proj.weight"}), p + "post_attention_layernorm.weight", D);
      const float scale = 1.4426950408889634f / sqrtf((float)dh);
      // (the merged query weight is only formed when its GEMM is asked for: MSH_XATTN_QT, or a shape the two-stage kernel
      // does not cover -- it is 8 D^3 multiply-adds per layer on the host)
      const bool need_wqk = !crossq2_supported(D, Hn) || xattn_qt_mode() != 0;
      std::vector<float> wqk(need_wqk ? (size_t)Hn * D * D : 0, 0.f), wvo((size_t)D * Hn * D, 0.f);
      const float* Wk = kv.data();
      const float* Wv = kv.data() + (size_t)D * D;
      msh_host::parallel_for((size_t)Hn, [&](size_t h) {
        for (int j = 0; need_wqk && j < dh; ++j) {
          const float* wkrow = Wk + (size_t)(h * dh + j) * D;
          const float* wqrow = wqf.data() + (size_t)(h * dh + j) * D;
          for (int d = 0; d < D; ++d) {
            const float a = wkrow[d] * scale;
            float* o = wqk.data() + ((size_t)h * D + d) * D;
            for (int k = 0; k < D; ++k) o[k] += a * wqrow[k];
          }
        }
        for (int n = 0; n < D; ++n) {
          float* o = wvo.data() + (size_t)n * Hn * D + h * D;
          for (int j = 0; j < dh; ++j) {
            const float a = wo_c[(size_t)n * D + h * dh + j];
            const float* wvrow = Wv + (size_t)(h * dh + j) * D;
            for (int d = 0; d < D; ++d) o[d] += a * wvrow[d];
          }
        }
      });
      if (need_wqk) upload_bf16_fm(wqk, Hn * D, D, &L.wqk);
      upload_bf16_fm(wvo, D, Hn * D, &L.wvo);
      if (crossq2_supported(D, Hn)) {   // the factors of wqk, kept apart (k_crossq.hip)
        std::vector<float> w1((size_t)Hn * 64 * D, 0.f);
        for (int h = 0; h < Hn; ++h)
          for (int j = 0; j < dh; ++j)
            for (int k = 0; k < D; ++k) w1[((size_t)h * 64 + j) * D + k] = scale * wqf[(size_t)(h * dh + j) * D + k];
        upload_bf16_fm(w1, Hn * 64, D, &L.wq1);
        std::vector<bf16_t> w2((size_t)Hn * (D / 16) * 2 * 64 * 8);
        pack_crossq_wk(Wk, D, Hn, w2.data());
        void* dp = weight_alloc(w2.size() * sizeof(bf16_t));
        copy_blocking(dp, w2.data(), w2.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
        L.wk2 = reinterpret_cast<bf16_t*>(dp);
      }
    }
    expect_shape(p + "mlp.fc1.weight", {2 * F, D});
    expect_shape(p + "mlp.fc2.weight", {D, F});
    std::vector<float> w = st.to_f32(p + "mlp.fc1.weight"), b = vec(p + "mlp.fc1.bias", 2 * F);
    std::vector<float> wi((size_t)2 * F * D), bi((size_t)2 * F);
    for (int j = 0; j < F; ++j) {  // first half = value, second half = gate (modeling_moonshine.py:92-96)
      memcpy(&wi[(size_t)(2 * j) * D], &w[(size_t)j * D], D * sizeof(float));
      memcpy(&wi[(size_t)(2 * j + 1) * D], &w[(size_t)(F + j) * D], D * sizeof(float));
      bi[2 * j] = b[j];
      bi[2 * j + 1] = b[F + j];
    }
    upload_bf16_fm(fold(wi, p + "final_layernorm.weight", 2 * F), 2 * F, D, &L.fc1);
    upload(bi, &L.b1);
    upload_bf16_fm(st.to_f32(p + "mlp.fc2.weight"), D, F, &L.fc2);
    upload(vec(p + "mlp.fc2.bias", D), &L.b2);
    upload(vec(p + "input_layernorm.weight", D), &L.ln1);
    upload(vec(p + "post_attention_layernorm.weight", D), &L.ln2);
    upload(vec(p + "final_layernorm.weight", D), &L.ln3);
  }
  upload_bf16(cross, &cross_kv_w_);
  if (const char* gap = dev_getenv("MSH_ALLOC_GAP_KB")) {   // developer probe: a dummy block between the weights and the workspaces
    const long kb = atol(gap);
    if (kb > 0 && !dry_run_) {
      std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
      weight_allocs_.push_back(device_alloc((size_t)kb << 10));
    }
  }
  // the same weight packed for the panel kernel (k_panel.hip): only when that instance is asked for (it is off by default,
  // see run_encoder)
  const char* ckv_load_env = dev_getenv("MSH_ENC_CROSS_KV_PANEL");
  if (cross_kv_panel_supported(D) && !dry_run_ && ckv_load_env != nullptr && ckv_load_env[0] == '2') {
    const int Lc = c.dec_layers;
    std::vector<bf16_t> packed(panel_packed_elems(Lc * 2 * D, D));
    pack_panel_weights(cross.data(), nullptr, Lc * 2 * D, D, packed.data());
    void* dp = weight_alloc(packed.size() * sizeof(bf16_t));
    copy_blocking(dp, packed.data(), packed.size() * sizeof(bf16_t), hipMemcpyHostToDevice);
    cross_kv_panel_w_ = reinterpret_cast<bf16_t*>(dp);
  }
  {
    // Scales of the optional fp8 (e4m3) cross K / V (set_kv_dtype): column n of the cross-KV GEMM is w_n . enc with
    // enc = LayerNorm(x) * gamma, so |value| <= |w_n|_2 * sqrt(D) * max|gamma| whatever the audio (Cauchy-Schwarz on a
    // normalised row).  qscale maps that bound to 448, e4m3's largest value: nothing can overflow, typical values
    // (~ bound / sqrt(D)) land around 20, three binades of normal range below them.  dq = 1 / qscale is what the decode
    // kernel applies to the query (K) and to the output (V).
    const std::vector<float> genc = vec("model.encoder.layer_norm.weight", D);
    float gmax = 0.f;
    for (float g : genc) gmax = std::max(gmax, fabsf(g));
    const size_t N = (size_t)c.dec_layers * 2 * D;
    std::vector<float> qs(N), dq(N);
    for (size_t n = 0; n < N; ++n) {
      double ss = 0.0;
      for (int k = 0; k < D; ++k) {
        const double w = (double)bf16_to_f32(f32_to_bf16(cross[n * D + k]));
        ss += w * w;
      }
      const float bound = (float)(sqrt(ss) * sqrt((double)D)) * gmax;
      qs[n] = bound > 0.f ? 448.0f / bound : 1.0f;
      dq[n] = 1.0f / qs[n];
    }
    upload(qs, &kv_qscale_);
    upload(dq, &kv_dq_);
  }

  // RoPE tables in fp32, computed the way the float definition does (modeling_moonshine.py:132-154):
  // inv_freq = 1 / theta^(2j/dim), angle = pos * inv_freq, then cos / sin.
  rope_max_pos_ = 8192;
  const int rp = cfg_.rot_pairs(), dim = rp * 2;
  std::vector<float> cs((size_t)rope_max_pos_ * rp), sn((size_t)rope_max_pos_ * rp);
  for (int j = 0; j < rp; ++j) {
    const float inv = 1.0f / powf(cfg_.rope_theta, (float)(2 * j) / (float)dim);
    for (int pos = 0; pos < rope_max_pos_; ++pos) {
      const float a = (float)pos * inv;
      cs[(size_t)pos * rp + j] = cosf(a);
      sn[(size_t)pos * rp + j] = sinf(a);
    }
  }
  upload(cs, &rope_cos_);
  upload(sn, &rope_sin_);
  // The LM head is tied to the embedding (configuration_moonshine.py:103; the engine multiplies by embed_tokens).  A
  // checkpoint may carry the alias `proj_out.weight` as well: accepted when it IS the embedding, refused when it is not --
  // an untied head would otherwise be silently ignored.
  if (st.has("proj_out.weight")) {
    expect_shape("proj_out.weight", {V, D});
    const std::vector<float> head = st.to_f32("proj_out.weight"), emb = st.to_f32("model.decoder.embed_tokens.weight");
    for (size_t i = 0; i < head.size(); ++i)
      if (head[i] != emb[i])
        throw std::runtime_error("proj_out.weight differs from model.decoder.embed_tokens.weight: this engine implements the tied "
                                 "LM head of the Moonshine configuration (tie_word_embeddings)");
  }
  // Anything else in the file is something this loader does not understand: say so instead of transcribing with half a
  // model.  (Position buffers some exporters persist are the only benign extras known.)
  {
    std::string extra;
    int n_extra = 0;
    for (const std::string& name : st.unused()) {
      if (name.size() >= 8 && name.compare(name.size() - 8, 8, "inv_freq") == 0) continue;
      if (n_extra++ < 6) extra += (extra.empty() ? "" : ", ") + name;
    }
    if (n_extra > 0)
      throw std::runtime_error(std::to_string(n_extra) + " tensor(s) in the checkpoint are not part of the Moonshine " + c.arch +
                               " architecture this engine loads: " + extra + (n_extra > 6 ? ", ..." : ""));
  }
  loaded_ = true;
}

// ------------------------------------------------------------------------------------------------
// Profiling scopes (HIP events on the engine stream)
// ------------------------------------------------------------------------------------------------
hipEvent_t Engine::get_event() {
  if (!event_pool_.empty()) {
    hipEvent_t e = event_pool_.back();
    event_pool_.pop_back();
    return e;
  }
  hipEvent_t e;
  MSH_HIP(hipEventCreate(&e));
  return e;
}

struct Engine::ProfScope {
  Engine* e;
  int idx = -1;
  hipEvent_t a{}, b{};
  ProfScope(Engine* eng, const char* name, double flops, double bytes) : e(eng) {
    if (!e->prof_on_) return;
    auto it = e->prof_idx_.find(name);
    if (it == e->prof_idx_.end()) {
      idx = (int)e->prof_.size();
      e->prof_idx_[name] = idx;
      ProfEntry pe;
      pe.name = name;
      e->prof_.push_back(pe);
    } else {
      idx = it->second;
    }
    e->prof_[idx].flops += flops;
    e->prof_[idx].bytes += bytes;
    e->prof_[idx].launches += 1;
    a = e->get_event();
    b = e->get_event();
    MSH_HIP(hipEventRecord(a, e->stream_));
  }
  ~ProfScope() {
    if (idx < 0) return;
    (void)hipEventRecord(b, e->stream_);
    e->prof_pending_.push_back({idx, a, b});
  }
};

double Engine::profile_event_overhead_ms(int iters) {
  MSH_HIP(hipSetDevice(device_));
  if (iters <= 0) iters = 1;
  std::vector<hipEvent_t> ev(2 * (size_t)iters);
  for (auto& e : ev) e = get_event();
  for (int i = 0; i < iters; ++i) {
    MSH_HIP(hipEventRecord(ev[2 * i], stream_));
    MSH_HIP(hipEventRecord(ev[2 * i + 1], stream_));
  }
  MSH_HIP(hipStreamSynchronize(stream_));
  double total = 0;
  for (int i = 0; i < iters; ++i) {
    float ms = 0.f;
    MSH_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    total += ms;
  }
  for (auto& e : ev) event_pool_.push_back(e);
  return total / iters;
}

void Engine::prof_flush() {
  if (prof_pending_.empty()) return;
  MSH_HIP(hipStreamSynchronize(stream_));
  for (auto& r : prof_pending_) {
    float ms = 0.f;
    MSH_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    prof_[r.idx].ms += ms;
    event_pool_.push_back(r.a);
    event_pool_.push_back(r.b);
  }
  prof_pending_.clear();
}

void Engine::profile_enable(bool on) {
  prof_flush();
  prof_on_ = on;
}
void Engine::profile_reset() {
  prof_flush();
  prof_.clear();
  prof_idx_.clear();
}
std::vector<ProfEntry> Engine::profile_get() {
  prof_flush();
  return prof_;
}

// ------------------------------------------------------------------------------------------------
// Batch planning
// ------------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

void Engine::plan_batch(const uint64_t* n_samples, uint32_t count, float mtps) {
  if (count == 0) throw std::invalid_argument("empty batch");
  // the clip index is a grid y / z coordinate of the stem and attention launches (k_misc.hip, k_attn.hip)
  if (count > 65535) throw std::invalid_argument("batch of " + std::to_string(count) + " clips: at most 65535 per call");
  clips_h_.assign(count, ClipMeta{});
  long row = 0, kv = 0;
  max_rows_ = 0;
  max_steps_ = 0;
  for (uint32_t i = 0; i < count; ++i) {
    const uint64_t n = n_samples[i];
    if (n > (1u << 30)) throw std::invalid_argument("clip too long");
    ClipMeta& c = clips_h_[i];
    // conv length formula: transformers modeling_moonshine.py:500-508
    const int L1 = n >= 127 ? (int)((n - 127) / 64 + 1) : 0;
    const int L2 = L1 >= 7 ? (L1 - 7) / 3 + 1 : 0;
    const int T = L2 >= 3 ? (L2 - 3) / 2 + 1 : 0;
    if (T < 1) throw std::invalid_argument("clip " + std::to_string(i) + " is too short (" + std::to_string(n) +
                                           " samples): the conv stem needs at least 895");
    c.n_samples = (int)n;
    c.L1 = L1;
    c.L2 = L2;
    c.T = T;
    // rows: the clip's slot in every stream.  conv1 consumes samples [0, 64*L1 + 63) from a slot of
    // 384*rows samples; conv2/conv3 need 2*rows >= L2 and rows >= T.
    int rows = std::max({(64 * L1 + 63 + 383) / 384, (L2 + 1) / 2, T});
    c.rows = round_up(rows, 8);  // 8: a clip starts on an 8-row boundary, so 8 consecutive keys never straddle clips
    c.row_start = (int)row;
    c.Tk = round_up(T, 8);
    c.kv_start = (int)kv;
    // step budget: reference core/moonshine-model.cpp:347-349 (float arithmetic)
    const float dur = (float)n / 16000.0f;
    c.max_len = (int)ceilf(dur * mtps);
    if (c.max_len < 1) c.max_len = 1;
    row += c.rows;
    kv += c.Tk;
    max_rows_ = std::max(max_rows_, c.rows);
    max_steps_ = std::max(max_steps_, c.max_len);
    if (T > rope_max_pos_) throw std::invalid_argument("clip too long for the RoPE table");
  }
  if (row > (1L << 24)) throw std::invalid_argument("batch too large");
  R_ = row;
  kv_keys_ = kv;
  n_clips_ = count;
}

// ------------------------------------------------------------------------------------------------
void Engine::encode(const float* const* pcm, const uint64_t* n_samples, uint32_t count, bool on_device, float mtps) {
  MSH_HIP(hipSetDevice(device_));
  if (!loaded_) throw std::runtime_error("no weights loaded");
  encoded_ = false;
  cross_counts_.clear();  // a new batch invalidates the captured attention of the previous one
  plan_batch(n_samples, count, mtps);
  const int D = cfg_.hidden, F = cfg_.ffn, L = cfg_.dec_layers;
  const long R = R_;
  bool moved = false;
  moved |= clips_d_.reserve(count * sizeof(ClipMeta));
  moved |= clip_ptrs_d_.reserve(count * sizeof(float*));
  moved |= audio_bf16_.reserve((384 * R + 512) * sizeof(bf16_t));
  moved |= row_pos_.reserve(R * sizeof(int));
  moved |= row_clip_.reserve(R * sizeof(int));
  moved |= x1_.reserve((6 * R + 16) * D * sizeof(bf16_t));
  moved |= x2_.reserve((2 * R + 8) * 2 * D * sizeof(bf16_t));
  moved |= H_.reserve(R * D * sizeof(float));
  const size_t Rp = (R + 127) / 128 * 128;   // the panel kernel (k_panel.hip) stores whole 128-row panels: rows past R are padding
  moved |= Y_.reserve(Rp * D * sizeof(bf16_t));   // (whole panels: the fused MLP hands the next layer's QKV its operand here)
  moved |= QKV_.reserve(Rp * 2 * D * sizeof(bf16_t));                // q | k rows of the current encoder layer
  moved |= VTe_.reserve(((size_t)D * Rp + 64) * sizeof(bf16_t));         // its V^T [D][Rp]: keys contiguous
  moved |= AO_.reserve(R * D * sizeof(bf16_t));
  moved |= Z_.reserve(R * F * sizeof(bf16_t));
  moved |= ENC_.reserve(R * D * sizeof(bf16_t));
  if (keep_enc_f32_) moved |= ENC32_.reserve(R * D * sizeof(float));
  moved |= gn_part_.reserve((size_t)count * 64 * sizeof(float2));
  moved |= gn_rows_.reserve((size_t)6 * R * gemm_max_col_tiles(D) * sizeof(float2));   // conv1's row sums (GroupNorm statistics)
  moved |= gn_stats_.reserve(count * sizeof(float2));
  moved |= gn_table_.reserve((size_t)count * 2 * D * sizeof(float));
  // Cross-attention form of this batch (k_xattn.hip): absorbed = the decode steps attend over ENC_ itself and no K^T / V^T
  // is projected.  One workgroup per clip: it needs a batch that fills the chip, the classic form (8 workgroups per clip)
  // stays for small batches, for the word-timestamp capture (which reads K^T) and for fp8 keys.
  // The form is a property of the ENGINE (set_cross_mode), never of the batch: the same clip decodes to the same ids whatever
  // shares its batch.  (MSH_XATTN_MIN_BATCH=n restores the round-4 per-batch switch of mode 0 for A/B measurements only.)
  if (cross_mode_ == 2 && (!cross_absorbed_available() || capture_cross_ || kv_fp8_))
    throw std::invalid_argument(std::string("cross_attention = absorbed cannot be honoured: ") +
                                (!cross_absorbed_available() ? "this architecture has no absorbed operands"
                                 : capture_cross_            ? "the word-timestamp capture reads the projected keys"
                                                             : "kv_dtype = fp8 stores projected keys"));
  absorbed_ = cross_mode_ == 2 || (cross_mode_ == 0 && xattn_min_batch() > 0 && cross_absorbed_available() && !capture_cross_ &&
                                   !kv_fp8_ && (int)count >= xattn_min_batch());
  if (!absorbed_) {
    moved |= KT_.reserve((size_t)L * D * kv_keys_ * kv_bytes());
    moved |= VT_.reserve((size_t)L * D * kv_keys_ * kv_bytes());
  }
  if (moved) ++ws_gen_;

  // clip pointers: stage host PCM into one device buffer, or use the caller's device pointers
  std::vector<const float*> ptrs(count);
  std::unique_lock<std::mutex> upload_turn;
  if (on_device) {
    for (uint32_t i = 0; i < count; ++i) ptrs[i] = pcm[i];
  } else {
    size_t total = 0;
    for (uint32_t i = 0; i < count; ++i) total += (n_samples[i] + 3) & ~size_t(3);
    pcm_stage_.reserve(total * sizeof(float));
    static const bool in_turn = [] {   // MSH_UPLOAD_IN_TURN=0: lanes upload side by side (A/B)
      const char* e = dev_getenv("MSH_UPLOAD_IN_TURN");
      return e == nullptr || atoi(e) != 0;
    }();
    if (in_turn && total * sizeof(float) >= ((size_t)8 << 20)) upload_turn = std::unique_lock<std::mutex>(device_upload_mutex(device_));
    std::vector<size_t> offs(count);
    size_t off = 0;
    for (uint32_t i = 0; i < count; ++i) {
      offs[i] = off;
      ptrs[i] = pcm_stage_.as<float>() + off;
      off += (n_samples[i] + 3) & ~size_t(3);
    }
    // A large batch in PAGEABLE host memory (what a caller of the C API hands over): hipMemcpyAsync would stage every clip
    // through the runtime's bounce buffers on this one thread (~10 GB/s, synchronous).  Gather the clips into this
    // engine's pinned buffer on a few host threads instead and move them with ONE asynchronous DMA.  Pinned or
    // registered caller memory, and small batches, are copied directly as before.
    bool pageable = false;
    if (total * sizeof(float) >= ((size_t)8 << 20)) {
      hipPointerAttribute_t at{};
      if (hipPointerGetAttributes(&at, pcm[0]) != hipSuccess) {
        (void)hipGetLastError();  // "not a registered pointer" is an answer here, not an error to keep
        pageable = true;
      } else {
        pageable = at.type == hipMemoryTypeUnregistered;
      }
    }
    if (pageable) {
      if (pcm_pinned_cap_ < total * sizeof(float)) {
        if (pcm_pinned_) MSH_HIP(hipHostFree(pcm_pinned_));
        pcm_pinned_ = nullptr;
        pcm_pinned_cap_ = 0;
        MSH_HIP(hipHostMalloc(&pcm_pinned_, total * sizeof(float), hipHostMallocDefault));
        pcm_pinned_cap_ = total * sizeof(float);
      }
      float* pin = static_cast<float*>(pcm_pinned_);
      msh_host::parallel_for(count, [&](size_t i) { memcpy(pin + offs[i], pcm[i], n_samples[i] * sizeof(float)); }, 8);
      MSH_HIP(hipMemcpyAsync(pcm_stage_.p, pin, total * sizeof(float), hipMemcpyHostToDevice, stream_));
    } else {
      for (uint32_t i = 0; i < count; ++i)
        MSH_HIP(hipMemcpyAsync(pcm_stage_.as<float>() + offs[i], pcm[i], n_samples[i] * sizeof(float), hipMemcpyHostToDevice, stream_));
    }
  }
  MSH_HIP(hipMemcpyAsync(clips_d_.p, clips_h_.data(), count * sizeof(ClipMeta), hipMemcpyHostToDevice, stream_));
  MSH_HIP(hipMemcpyAsync(clip_ptrs_d_.p, ptrs.data(), count * sizeof(float*), hipMemcpyHostToDevice, stream_));
  MSH_HIP(hipStreamSynchronize(stream_));  // `ptrs` / `clips_h_` staging is on the host stack
  if (upload_turn.owns_lock()) upload_turn.unlock();
  run_encoder();
  prof_flush();
  encoded_ = true;
}

void Engine::run_encoder() {
  const int D = cfg_.hidden, F = cfg_.ffn, Hh = cfg_.heads, L = cfg_.dec_layers;
  const int R = (int)R_;
  const ClipMeta* clips = clips_d_.as<ClipMeta>();
  hipStream_t s = stream_;
  // algorithmic work (valid frames only) for the profiler
  double sL1 = 0, sL2 = 0, sT = 0, sT2 = 0, sN = 0;
  for (const ClipMeta& c : clips_h_) {
    sL1 += c.L1;
    sL2 += c.L2;
    sT += c.T;
    sT2 += (double)c.T * c.T;
    sN += c.n_samples;
  }
  RopeParams rp{rope_cos_, rope_sin_, cfg_.rot_pairs(), cfg_.head_dim(), D};
  // MSH_ENC_MLP=0: the MLP block as LayerNorm + two tiled GEMMs (A/B switch; the fused kernel is the default)
  // developer switch (read per call, so that one test process can compare both paths): 0 = the tiled GEMMs, 2 = the panel
  // kernel at any batch size
  const char* qkv_env = dev_getenv("MSH_ENC_QKV_PANEL");
  const bool qkv_panel_on = !(qkv_env != nullptr && qkv_env[0] == '0');
  const long qkv_panel_min_rows = ((qkv_env != nullptr && qkv_env[0] == '2') || uniform_kernels_) ? 8 : 128 * 128;   // (uniform_kernels_: the large-batch kernels for every call)
  // developer switch (per call): 0 = tiled GEMMs, 2 = the MLP block alone in the fused kernel behind a tiled o-proj,
  // 3 = o-proj + MLP fused at any batch size; default 1 = o-proj + MLP fused from 32 k rows on
  const char* mlp_env = dev_getenv("MSH_ENC_MLP");
  const int fused_mlp = mlp_env == nullptr ? 1 : (mlp_env[0] == '0' ? 0 : mlp_env[0] == '2' ? 2 : 1);
  const long mlp_min_rows = ((mlp_env != nullptr && mlp_env[0] == '3') || uniform_kernels_) ? 1 : 128 * 256;
  // Layer l's fused o-proj + MLP kernel hands layer l + 1's QKV panel kernel its operand -- LayerNorm of the rows it just
  // produced, bf16, fragment-major (k_mlp.hip YOUT, k_panel.hip AM = 2) -- whenever both layers run on those kernels
  // (MSH_ENC_LN_HANDOVER=0: the panel kernel fetches and normalises the fp32 rows itself, as before round 6)
  const char* hand_env = dev_getenv("MSH_ENC_LN_HANDOVER");
  const bool hand_over_on = !(hand_env != nullptr && hand_env[0] == '0');
  bool y_handed = false;   // Y_ holds the current layer's normalised rows in fragment-major order

  {
    ProfScope p(this, "pack_audio", 0, sN * 4 + 384.0 * R * 2);
    pack_audio(clip_ptrs_d_.as<const float*>(), clips, (int)n_clips_, audio_bf16_.as<bf16_t>(), 0, s);
    build_row_meta(clips, (int)n_clips_, row_pos_.as<int>(), row_clip_.as<int>(), s);
  }
  const char* gn_env = dev_getenv("MSH_GN_ROWSUMS");   // (per call, like the switches above)
  const bool gn_rowsums = !(gn_env != nullptr && gn_env[0] == '0');
  int gn_ntn = 0;
  {  // conv1 (k127, s64, no bias) + tanh: GEMM over the sample stream, row t = samples [64t, 64t+128)
    ProfScope p(this, "conv1_tanh_gemm", 2.0 * sL1 * D * 127, sN * 2 + sL1 * D * 2);
    // (its epilogue leaves the row sums of what it stores: the GroupNorm statistics never read the 532 MB back;
    //  MSH_GN_ROWSUMS=0: the statistics pass over the output, as before round 6)
    gn_ntn = gemm_tanh_bf16(audio_bf16_.as<bf16_t>(), 64, conv1_w_, 6 * R, D, 128, x1_.as<bf16_t>(),
                            gn_rowsums ? gn_rows_.as<float2>() : nullptr, s);
  }
  {  // GroupNorm(1 group): only the per-clip statistics are computed; the affine map is folded into conv2
    ProfScope p(this, "groupnorm_stats", 0, gn_rowsums ? sL1 * gn_ntn * 8.0 : sL1 * D * 2);
    if (gn_rowsums) groupnorm_stats_rows(gn_rows_.as<float2>(), gn_ntn, clips, (int)n_clips_, D, gn_part_.as<float>(), gn_stats_.as<float2>(), s);
    else groupnorm_stats(x1_.as<bf16_t>(), clips, (int)n_clips_, D, gn_part_.as<float>(), gn_stats_.as<float2>(), s);
    gn_fold_table(gn_stats_.as<float2>(), conv2_s1_, conv2_b2_, (int)n_clips_, 2 * D, gn_table_.as<float>(), s);
  }
  {  // conv2 (k7, s3) + GroupNorm fold + GELU: window of 7 channels-last frames is contiguous -> lda = 3D, K = 7D
    ProfScope p(this, "conv2_gelu_gemm", 2.0 * sL2 * 2 * D * 7 * D, sL1 * D * 2 + sL2 * 2 * D * 2);
    gemm_gn_bias_gelu_bf16(x1_.as<bf16_t>(), 3L * D, conv2_w_, gn_table_.as<float>(), gn_stats_.as<float2>(),
                           row_clip_.as<int>(), 2 * R, 2 * D, 7 * D, x2_.as<bf16_t>(), conv_kperm_, s);
  }
  {  // conv3 (k3, s2) + GELU -> residual stream H [R, D] fp32
    ProfScope p(this, "conv3_gelu_gemm", 2.0 * sT * D * 6 * D, sL2 * 2 * D * 2 + sT * D * 4);
    gemm_bias_gelu_f32(x2_.as<bf16_t>(), 4L * D, conv3_w_, conv3_b_, R, D, 6 * D, H_.as<float>(), conv_kperm_, s);
  }
  // A few clips (the latency case; one 10 s clip = 424 rows): the tiled kernel's 128 x 208 tiles give a layer's GEMMs 8 .. 32
  // workgroups on 256 CUs, each walking K alone (fc2: 52 slices, 35 us at one clip).  The split-K decode GEMM (16 x 32 tiles,
  // four waves over K, operands straight from global memory) puts 350 .. 1400 workgroups on the same shapes: MSH_ENC_SMALL_ROWS
  // = largest row count that takes it (default 1024; 0 = off).  Same epilogue arithmetic, a different summation order over K.
  const char* small_env = dev_getenv("MSH_ENC_SMALL_ROWS");
  const long small_rows = small_env != nullptr ? atol(small_env) : 1024;
  const bool small_gemms = !uniform_kernels_ && R <= small_rows && (R & 3) == 0 && qkv_env == nullptr && mlp_env == nullptr;   // (a developer switch that names a kernel gets that kernel)
  // Two halves side by side.  A layer of a large batch is three chip-filling kernels whose grids end in a partly filled round
  // (848 panels of 128 rows on 256 CUs = 3.31 rounds for 256 x 10 s: the fused MLP takes the time of 4).  Every kernel of the
  // layer loop works row by row or clip by clip, so the batch is cut at a clip boundary on a panel boundary and the two halves
  // run as two chains on two streams: one half's kernels take the CUs the other half's last round leaves idle.  Same bits.
  // Only when the engine has the GPU to itself (serial 39.3 -> 38.6 ms per 256 x 10 s batch): with batches in flight the
  // other lanes already fill those CUs and the second stream is one more queue to arbitrate (93.7 -> 92.5-93.2 k audio-s/s).
  // MSH_ENC_SPLIT=0 / 1: never / also with lanes.
  const char* split_env = dev_getenv("MSH_ENC_SPLIT");
  const bool split_on = split_env != nullptr ? split_env[0] == '1' : !shared_gpu_;
  int c_split = -1;
  if (split_on && !prof_on_ && !small_gemms && qkv_panel_on && fused_mlp == 1 && (R & 7) == 0) {
    long best = -1;
    for (uint32_t c = 1; c < n_clips_; ++c) {
      const long rs = clips_h_[c].row_start;
      if (rs % 128 != 0) continue;
      if (best < 0 || std::labs(rs - R / 2) < std::labs(best - R / 2)) best = rs, c_split = (int)c;
    }
    const long need = std::max(qkv_panel_min_rows, mlp_min_rows);
    if (c_split > 0 && (best < need || R - best < need)) c_split = -1;
    for (int l = 0; l < cfg_.enc_layers && c_split > 0; ++l)
      if (enc_[l].qkv_panel == nullptr || enc_[l].mlp == nullptr) c_split = -1;
  }
  if (c_split > 0) {
    if (enc_stream2_ == nullptr) {
      MSH_HIP(hipStreamCreateWithFlags(&enc_stream2_, hipStreamNonBlocking));
      MSH_HIP(hipEventCreateWithFlags(&enc_fork_, hipEventDisableTiming));
      MSH_HIP(hipEventCreateWithFlags(&enc_join_, hipEventDisableTiming));
    }
    MSH_HIP(hipEventRecord(enc_fork_, s));
    MSH_HIP(hipStreamWaitEvent(enc_stream2_, enc_fork_, 0));
    const long vt_ld = (long)((R + 127) / 128 * 128);
    const long r_cut = clips_h_[c_split].row_start;
    const bool nt = enc_store_nt(shared_gpu_);
    for (int l = 0; l < cfg_.enc_layers; ++l) {
      const EncLayerW& W = enc_[l];
      for (int h = 0; h < 2; ++h) {
        hipStream_t hs = h == 0 ? s : enc_stream2_;
        const long r0 = h == 0 ? 0 : r_cut, Rn = h == 0 ? r_cut : R - r_cut;
        const int c0 = h == 0 ? 0 : c_split, nc = h == 0 ? c_split : (int)n_clips_ - c_split;
        float* Hh_ = H_.as<float>() + r0 * D;
        bf16_t* Yh = Y_.as<bf16_t>() + r0 * D;
        bf16_t* QKVh = QKV_.as<bf16_t>() + r0 * 2 * D;
        bf16_t* AOh = AO_.as<bf16_t>() + r0 * D;
        if (l > 0 && hand_over_on)
          qkv_panel_prenorm(Yh, W.qkv_panel, (int)Rn, D, row_pos_.as<int>() + r0, rp, QKVh, VTe_.as<bf16_t>() + r0, vt_ld, hs, nt);
        else
          qkv_panel(Hh_, W.qkv_panel, (int)Rn, D, row_pos_.as<int>() + r0, rp, QKVh, VTe_.as<bf16_t>() + r0, vt_ld, hs, nt);
        enc_attention(QKV_.as<bf16_t>(), VTe_.as<bf16_t>(), vt_ld, AO_.as<bf16_t>(), clips + c0, nc, max_rows_, D, Hh, hs);
        const bool hand = hand_over_on && l + 1 < cfg_.enc_layers;
        mlp_fused_oproj(Hh_, AOh, W.mlp, W.b2, Rn, D, F, hs, nt, hand ? Yh : nullptr);
      }
    }
    MSH_HIP(hipEventRecord(enc_join_, enc_stream2_));
    MSH_HIP(hipStreamWaitEvent(s, enc_join_, 0));
  }
  for (int l = 0; l < cfg_.enc_layers && c_split <= 0; ++l) {
    const EncLayerW& W = enc_[l];
    long vt_ld = (long)R;
    if (small_gemms) {
      bool ok = true;
      {
        ProfScope p(this, "enc_layernorm", 0, sT * D * 6);
        layernorm_bf16(H_.as<float>(), W.ln1, R, D, Y_.as<bf16_t>(), nullptr, s);
      }
      {
        ProfScope p(this, "enc_qkv_rope_gemm", 2.0 * sT * D * 3 * D, sT * D * 2 * 5);
        ok = ok && small_gemm_qkv_rope_bf16(Y_.as<bf16_t>(), D, W.wqkv, R, 2 * D, D, row_pos_.as<int>(), rp, QKV_.as<bf16_t>(), s);
        // V^T [D][R] = Wv x Y^T: the weight is the row operand, the stream rows are the output columns
        ok = ok && small_gemm_act(W.wqkv + (size_t)2 * D * D, D, Y_.as<bf16_t>(), nullptr, 0, D, (int)R, D, VTe_.as<bf16_t>(), nullptr, s);
      }
      {
        ProfScope p(this, "enc_attention", 4.0 * sT2 * D, sT * D * 2 * 4);
        enc_attention(QKV_.as<bf16_t>(), VTe_.as<bf16_t>(), vt_ld, AO_.as<bf16_t>(), clips, (int)n_clips_, max_rows_, D, Hh, s);
      }
      {
        ProfScope p(this, "enc_oproj_gemm", 2.0 * sT * D * D, sT * D * (2 + 8));
        ok = ok && small_gemm_resid_f32(AO_.as<bf16_t>(), D, W.wo, nullptr, R, D, D, H_.as<float>(), s);
      }
      {
        ProfScope p(this, "enc_layernorm", 0, sT * D * 6);
        layernorm_bf16(H_.as<float>(), W.ln2, R, D, Y_.as<bf16_t>(), nullptr, s);
      }
      {
        ProfScope p(this, "enc_fc1_gelu_gemm", 2.0 * sT * D * F, sT * (D + F) * 2);
        ok = ok && small_gemm_act(Y_.as<bf16_t>(), D, W.fc1, W.b1, 2, R, F, D, Z_.as<bf16_t>(), nullptr, s);
      }
      {
        ProfScope p(this, "enc_fc2_gemm", 2.0 * sT * D * F, sT * (F * 2 + D * 8));
        ok = ok && small_gemm_resid_f32(Z_.as<bf16_t>(), F, W.fc2, W.b2, R, D, F, H_.as<float>(), s);
      }
      if (!ok) throw std::logic_error("run_encoder: a width of this architecture is not compiled into the split-K GEMM");
      continue;
    }
    if (qkv_panel_on && W.qkv_panel != nullptr && R >= qkv_panel_min_rows && (R & 7) == 0) {
      // LayerNorm + q | k with RoPE + V transposed in one A-stationary panel kernel (k_panel.hip); below about half a
      // panel per CU the tiled GEMMs, whose tiles are smaller, fill the chip better
      ProfScope p(this, "enc_qkv_panel", 2.0 * sT * D * 3 * D, sT * D * (4 + 6));
      vt_ld = (long)((R + 127) / 128 * 128);
      const bool take = y_handed;
      y_handed = false;
      if (take)
        qkv_panel_prenorm(Y_.as<bf16_t>(), W.qkv_panel, (int)R, D, row_pos_.as<int>(), rp, QKV_.as<bf16_t>(), VTe_.as<bf16_t>(), vt_ld, s, enc_store_nt(shared_gpu_));
      else
        qkv_panel(H_.as<float>(), W.qkv_panel, (int)R, D, row_pos_.as<int>(), rp, QKV_.as<bf16_t>(), VTe_.as<bf16_t>(), vt_ld, s, enc_store_nt(shared_gpu_));
    } else {
      {
        ProfScope p(this, "enc_layernorm", 0, sT * D * 6);
        layernorm_bf16(H_.as<float>(), W.ln1, R, D, Y_.as<bf16_t>(), nullptr, s);
      }
      {  // q | k with RoPE, row-major [R][2D]
        ProfScope p(this, "enc_qkv_rope_gemm", 2.0 * sT * D * 2 * D, sT * D * 2 * 3);
        gemm_qkv_rope_bf16(Y_.as<bf16_t>(), D, W.wqkv, R, 2 * D, D, row_pos_.as<int>(), rp, QKV_.as<bf16_t>(), s);
      }
      {  // v TRANSPOSED, as one plain GEMM with the operands swapped: V^T [D][R] = Wv [D][D] x Y^T -- the weight is the
         // "activation" (4 row tiles), the R stream rows are the output columns, so the keys of a clip are contiguous in
         // every row of the result: what the attention kernel's P.V wants as its MFMA operand
        ProfScope p(this, "enc_qkv_rope_gemm", 2.0 * sT * D * D, sT * D * 2 * 2);
        gemm_act(W.wqkv + (size_t)2 * D * D, D, Y_.as<bf16_t>(), nullptr, 0, D, (int)R, D, VTe_.as<bf16_t>(), nullptr, s);
      }
    }
    {
      ProfScope p(this, "enc_attention", 4.0 * sT2 * D, sT * D * 2 * 4);
      enc_attention(QKV_.as<bf16_t>(), VTe_.as<bf16_t>(), vt_ld, AO_.as<bf16_t>(), clips, (int)n_clips_, max_rows_, D, Hh, s);
    }
    // (a panel is 128 rows and a CU holds one: below ~one panel per CU the tiled GEMMs, whose tiles are 8x smaller, fill
    // the chip better -- 13,568 rows: fused 0.104 ms, tiled 0.082 + LayerNorm; 32,768 rows: 0.124 against 0.16)
    const bool mlp_fused_now = fused_mlp != 0 && W.mlp != nullptr && R >= mlp_min_rows;
    if (mlp_fused_now && fused_mlp == 1) {
      // o-proj + residual + LayerNorm + fc1 + GELU + fc2 + residual in ONE kernel: H is read once and written once per layer
      // for both blocks, the [R][F] intermediate never exists (k_mlp.hip)
      ProfScope p(this, "enc_oproj_mlp_fused", 2.0 * sT * D * D + 4.0 * sT * D * F, sT * D * (2 + 8));
      // (the next layer takes the hand-over iff it runs the panel kernel: the same test as at the top of this loop)
      const bool next_panel = l + 1 < cfg_.enc_layers && qkv_panel_on && enc_[l + 1].qkv_panel != nullptr && R >= qkv_panel_min_rows && (R & 7) == 0;
      y_handed = hand_over_on && next_panel;
      mlp_fused_oproj(H_.as<float>(), AO_.as<bf16_t>(), W.mlp, W.b2, R, D, F, s, enc_store_nt(shared_gpu_), y_handed ? Y_.as<bf16_t>() : nullptr);
      continue;
    }
    {
      ProfScope p(this, "enc_oproj_gemm", 2.0 * sT * D * D, sT * D * (2 + 8));
      gemm_resid_f32(AO_.as<bf16_t>(), D, W.wo, nullptr, R, D, D, H_.as<float>(), s);
    }
    if (mlp_fused_now) {   // MSH_ENC_MLP=2: the MLP block alone in the fused kernel (its stages follow the o-proj ones)
      ProfScope p(this, "enc_mlp_fused", 4.0 * sT * D * F, sT * D * 8);
      mlp_fused(H_.as<float>(), W.mlp + (size_t)((D / 32 + 1) / 2) * (D / 8 + 1) * 512, W.b2, R, D, F, s);
      continue;
    }
    {
      ProfScope p(this, "enc_layernorm", 0, sT * D * 6);
      layernorm_bf16(H_.as<float>(), W.ln2, R, D, Y_.as<bf16_t>(), nullptr, s);
    }
    {
      ProfScope p(this, "enc_fc1_gelu_gemm", 2.0 * sT * D * F, sT * (D + F) * 2);
      gemm_bias_gelu_bf16(Y_.as<bf16_t>(), D, W.fc1, W.b1, R, F, D, Z_.as<bf16_t>(), s);
    }
    {
      ProfScope p(this, "enc_fc2_gemm", 2.0 * sT * D * F, sT * (F * 2 + D * 8));
      gemm_resid_f32(Z_.as<bf16_t>(), F, W.fc2, W.b2, R, D, F, H_.as<float>(), s);
    }
  }
  {
    ProfScope p(this, "enc_layernorm", 0, sT * D * 6);
    layernorm_bf16(H_.as<float>(), enc_ln_, R, D, ENC_.as<bf16_t>(), keep_enc_f32_ ? ENC32_.as<float>() : nullptr, s);
  }
  if (!absorbed_) {  // cross-attention K/V of all decoder layers in one GEMM, written as K^T / V^T for the decode stream
    // The panel kernel's cross-KV instance is OFF unless asked for (MSH_ENC_CROSS_KV_PANEL=2): at 256 x 10 s it measured
    // 1.01 ms against 0.93 ms for the A-stationary tiled kernel -- this GEMM writes 1.42 GB of K^T / V^T per batch and is
    // bound by that, not by its operand traffic.  Its parity test keeps the instance honest.
    const char* ckv_env = dev_getenv("MSH_ENC_CROSS_KV_PANEL");
    const bool ckv_panel = cross_kv_panel_w_ != nullptr && ckv_env != nullptr && ckv_env[0] == '2';
    ProfScope p(this, ckv_panel ? "cross_kv_panel" : "cross_kv_gemm", 2.0 * sT * D * 2 * D * L, sT * D * 2 + sT * D * 2.0 * L * 2);
    if (ckv_panel)
      cross_kv_panel(ENC_.as<bf16_t>(), cross_kv_panel_w_, (int)R, D, L, row_clip_.as<int>(), clips, (long)D * kv_keys_,
                     kv_fp8_ ? kv_qscale_ : nullptr, KT_.p, VT_.p, s);
    else if (kv_fp8_)
      gemm_cross_kv_fp8(ENC_.as<bf16_t>(), D, cross_kv_w_, R, L * 2 * D, D, row_clip_.as<int>(), clips, D, (long)D * kv_keys_,
                        kv_qscale_, KT_.as<uint8_t>(), VT_.as<uint8_t>(), s);
    else
      gemm_cross_kv(ENC_.as<bf16_t>(), D, cross_kv_w_, R, L * 2 * D, D, row_clip_.as<int>(), clips, D,
                    (long)D * kv_keys_, KT_.as<bf16_t>(), VT_.as<bf16_t>(), s);
  }
}

void Engine::get_encoder_output(uint32_t clip, float* out) {
  MSH_HIP(hipSetDevice(device_));
  if (!encoded_ || !keep_enc_f32_) throw std::runtime_error("encoder output not available (enable keep_encoder_f32)");
  const ClipMeta& c = clips_h_.at(clip);
  MSH_HIP(hipStreamSynchronize(stream_));
  copy_blocking(out, ENC32_.as<float>() + (long)c.row_start * cfg_.hidden,
                    (size_t)c.T * cfg_.hidden * sizeof(float), hipMemcpyDeviceToHost);
}

// ------------------------------------------------------------------------------------------------
// Decode
//
// The batch is decoded in lock-step, split into `groups` contiguous sub-batches that run on their own
// HIP streams: a decode step is a chain of short, latency-bound kernels (fused LN+GEMMs, self-attention)
// around one HBM-bound stream (cross-attention over K^T/V^T); with two or more independent chains in
// flight the short kernels of one group fill the CUs while another group streams its cross K/V.
// Each group's step (8 kernels per layer + LM head + bookkeeping) is captured once into a hipGraph and
// replayed: everything step-dependent (position, ids, masks) lives in device memory.
// ------------------------------------------------------------------------------------------------
struct Engine::DecodeGroup {
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int first = 0, M = 0;
  DevBuf dH, dq, dao, dz, dy, logits, cacheK, cacheV, tokens, counts, finished, scalars, teacher, pval, pidx, xpart;
  bool self_fused = false;    // single-clip latency path: self-attention inside the output projection's launch (<= 2 clips)
  bool loop_cross = false;    // 5 .. 63 clips: the same cross-attention with one workgroup per (clip, head) walking the slices
  bool split_cross = false;   // single-clip latency path: cross-attention split over key slices (k_dec_small.hip)
  int xs_max = 0;             // slices of the longest clip of the group
  bool fused_argmax = false;  // LM head writes per-tile (max, index) pairs instead of logits
  int argmax_tiles = 0;       // pairs per row: gemm_argmax_tiles(V)
  DecodeState state{};        // of the last decode() (profile_decode_chain replays its kernels)
  bool has_state = false;
  // Captured decode steps, a small LRU by shape key (everything baked into the kernel arguments): real batch calls are
  // ragged -- clip count, frames and step budget change from call to call -- and a one-entry cache re-captured on almost
  // every batch.  `g1` = one step, `gn` = graph_steps() consecutive steps in one replay (the loop's steady state); gn is
  // ~9x the nodes, so only the first shape an engine sees (fixed-shape workloads) and shapes that come back get one.
  struct GraphEntry {
    std::string key;
    hipGraphExec_t g1 = nullptr, gn = nullptr;
    uint64_t last = 0;
    uint32_t uses = 0;
  };
  static constexpr size_t kGraphCache = 8;
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0, graphs_ws_gen = 0, graphs_gen = 0;
  uint64_t captures = 0;              // graphs instantiated so far (msh_debug counter: tests assert the cache works)
  hipGraphExec_t graph = nullptr;     // of the current batch: one decode step (owned by `graphs`)
  hipGraphExec_t graph_n = nullptr;   // of the current batch: graph_steps() steps, or null (owned by `graphs`)
  uint64_t gen = 0;
  int32_t n_active_h = 0;
  void drop_graphs() {
    for (GraphEntry& ge : graphs) {
      if (ge.g1) (void)hipGraphExecDestroy(ge.g1);
      if (ge.gn) (void)hipGraphExecDestroy(ge.gn);
    }
    graphs.clear();
    graph = graph_n = nullptr;
  }
  ~DecodeGroup() {
    drop_graphs();
    DevBuf* bufs[] = {&dH, &dq, &dao, &dz, &dy, &logits, &cacheK, &cacheV, &tokens, &counts, &finished, &scalars, &teacher,
                      &pval, &pidx, &xpart};
    for (DevBuf* b : bufs) b->release();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

void Engine::destroy_groups() { groups_.clear(); }

double Engine::profile_cross_attention_ms(int rounds) {
  MSH_HIP(hipSetDevice(device_));
  if (!encoded_ || groups_.empty() || groups_[0]->M != (int)n_clips_)
    throw std::runtime_error("profile_cross_attention: encode and decode a batch first");
  // rounds < 0 (probe): every launch reads layer 0's K / V again -- one layer of a 256-clip batch is 177 MB, inside the
  // 256 MB memory-side cache, so this is the kernel's rate when its operands do NOT come from HBM
  const bool same_layer = rounds < 0;
  if (rounds < 0) rounds = -rounds;
  if (rounds == 0) rounds = 1;
  DecodeGroup& g = *groups_[0];
  const int D = cfg_.hidden, L = cfg_.dec_layers;
  MSH_HIP(hipStreamSynchronize(g.stream));
  hipEvent_t a = get_event(), b = get_event();
  auto sweep = [&] {
    for (int l = 0; l < L; ++l) {
      const int ll = same_layer ? 0 : l;
      if (absorbed_) {   // every layer reads the same encoder output (that IS the decode step's access pattern)
        dec_cross_absorbed(g.dq.as<bf16_t>(), ENC_.as<bf16_t>(), clips_d_.as<ClipMeta>(), g.M, D, cfg_.heads, g.dao.as<bf16_t>(), stream_);
        continue;
      }
      dec_cross_attention(g.dq.as<float>(), kv_layer(KT_, ll), kv_layer(VT_, ll), clips_d_.as<ClipMeta>(), g.M, D, cfg_.heads,
                          g.dao.as<bf16_t>(), stream_, kdq(ll), vdq(ll));
    }
  };
  sweep();  // warm
  MSH_HIP(hipEventRecord(a, stream_));
  for (int r = 0; r < rounds; ++r) sweep();
  MSH_HIP(hipEventRecord(b, stream_));
  MSH_HIP(hipStreamSynchronize(stream_));
  float ms = 0.f;
  MSH_HIP(hipEventElapsedTime(&ms, a, b));
  event_pool_.push_back(a);
  event_pool_.push_back(b);
  return (double)ms / ((double)rounds * L);
}


size_t Engine::debug_read(const std::string& name, void* dst, size_t bytes) {
  MSH_HIP(hipSetDevice(device_));
  if (name == "cross_k" || name == "cross_v") {   // K^T / V^T of the last encode(): [layers][D * keys] at kv_bytes() per key
    if (absorbed_) throw std::runtime_error("debug_read: the last batch used the absorbed cross-attention, K^T / V^T were not written");
    const size_t size = (size_t)cfg_.dec_layers * cfg_.hidden * kv_keys_ * kv_bytes();
    MSH_HIP(hipStreamSynchronize(stream_));
    if (dst != nullptr && bytes > 0) copy_blocking(dst, name == "cross_k" ? KT_.p : VT_.p, std::min(bytes, size), hipMemcpyDeviceToHost);
    return size;
  }
  if (name == "gn_stats") {   // GroupNorm {mean, rstd} per clip of the last encode()
    const size_t size = (size_t)n_clips_ * sizeof(float2);
    MSH_HIP(hipStreamSynchronize(stream_));
    if (dst != nullptr && bytes > 0) copy_blocking(dst, gn_stats_.p, std::min(bytes, size), hipMemcpyDeviceToHost);
    return size;
  }
  if (name == "graph_captures") {   // decode-step graphs instantiated so far (the LRU of DecodeGroup::graphs at work)
    size_t n = 0;
    for (const auto& g : groups_) n += (size_t)g->captures;
    return n;
  }
  if (groups_.empty()) throw std::runtime_error("debug_read: no decode() yet");
  DecodeGroup& g = *groups_[0];
  const int dh = cfg_.head_dim();
  const size_t cache = (size_t)cfg_.dec_layers * g.M * cfg_.heads * Smax_ * dh * sizeof(bf16_t);
  const void* src = nullptr;
  size_t size = 0;
  if (name == "cache_k") src = g.cacheK.p, size = cache;
  else if (name == "cache_v") src = g.cacheV.p, size = cache;
  else if (name == "resid") src = g.dH.p, size = (size_t)g.M * cfg_.hidden * sizeof(float);
  else throw std::invalid_argument("debug_read: unknown buffer " + name);
  MSH_HIP(hipStreamSynchronize(g.stream));
  if (dst != nullptr && bytes > 0) {
    if (name == "resid") {  // the device keeps the residual stream fragment-major (kernels.h fm32): hand back [clips][hidden]
      const int D = cfg_.hidden;
      std::vector<float> fm((size_t)round_up(g.M, 16) * D), rm((size_t)g.M * D);
      copy_blocking(fm.data(), src, fm.size() * sizeof(float), hipMemcpyDeviceToHost);
      for (int m = 0; m < g.M; ++m)
        for (int k = 0; k < D; ++k) rm[(size_t)m * D + k] = fm[(size_t)fm32(m, k, D >> 5)];
      memcpy(dst, rm.data(), std::min(bytes, size));
    } else {
      copy_blocking(dst, src, std::min(bytes, size), hipMemcpyDeviceToHost);
    }
  }
  return size;
}

void Engine::decode_step_enqueue(DecodeGroup& g) {
  const int D = cfg_.hidden, F = cfg_.ffn, Hh = cfg_.heads, V = cfg_.vocab, dh = cfg_.head_dim();
  const int M = g.M;
  hipStream_t s = g.stream;
  RopeParams rp{rope_cos_, rope_sin_, cfg_.rot_pairs(), dh, D};
  int32_t* pos = g.scalars.as<int32_t>();  // [0] = pos, [1] = n_active
  float* dH = g.dH.as<float>();
  float* dq = g.dq.as<float>();
  bf16_t* dao = g.dao.as<bf16_t>();
  bf16_t* dz = g.dz.as<bf16_t>();
  const ClipMeta* clips = clips_d_.as<ClipMeta>() + g.first;
  const size_t cache_layer = (size_t)M * Hh * Smax_ * dh;
  double sT = 0;
  for (int b = 0; b < M; ++b) sT += clips_h_[g.first + b].T;
  const double w_dd = 2.0 * D * D;  // bytes of a [D, D] bf16 weight
  dec_gemm_prefer_throughput(shared_gpu_ && M >= 192);
  // chain profiling (profile_decode_chain): enqueue only the kernel group `step_only_`
  auto on = [&](int id) { return step_mask_ != 0 ? ((step_mask_ >> id) & 1u) != 0 : (step_only_ < 0 || step_only_ == id); };
  for (int l = 0; l < cfg_.dec_layers; ++l) {
    const DecLayerW& W = dec_[l];
    bf16_t* cK = g.cacheK.as<bf16_t>() + l * cache_layer;
    bf16_t* cV = g.cacheV.as<bf16_t>() + l * cache_layer;
    if (on(0)) {
      ProfScope p(this, "dec_qkv_gemm", 2.0 * M * D * 3 * D, 3 * w_dd + M * D * 4.0 * 2);
      dec_gemm_qkv(dH, W.wqkv, M, D, pos, rp, dq, cK, cV, Smax_, s);
    }
    if (g.self_fused) {
      // single-clip latency path (k_dec_small.hip): self-attention + output projection + residual in one launch
      if (on(1)) {
        ProfScope p(this, "dec_self_attention", 2.0 * M * D * D, w_dd + M * D * 14.0 + 2.0 * M * D * 2.0 * 33);
        dec_self_oproj(dq, cK, cV, pos, W.wo, M, D, Hh, Smax_, dH, s);
      }
    } else {
      if (on(1)) {
        // q (fp32) + the cached K / V rows of every (clip, head) up to the current position + the bf16 output
        ProfScope p(this, "dec_self_attention", 0, M * D * 6.0 + 2.0 * M * D * 2.0 * 33);
        dec_self_attention(dq, cK, cV, pos, M, D, Hh, Smax_, dao, s);
      }
      if (on(2)) {
        ProfScope p(this, "dec_proj_resid_gemm", 2.0 * M * D * D, w_dd + M * D * 10.0);
        dec_gemm_resid(dao, W.wo, nullptr, M, D, D, dH, s);
      }
    }
    static const bool fuse_q = [] {
      const char* e = dev_getenv("MSH_NO_FUSED_CROSSQ");
      return !(e != nullptr && e[0] == '1');
    }();
    const bf16_t* KTl = kv_layer(KT_, l);
    const bf16_t* VTl = kv_layer(VT_, l);
    if (absorbed_) {
      // k_xattn.hip: qt = LN(h) Wqk^T (all heads' keys-side queries, D wide each), one pass over the encoder output for the
      // attention of all heads, then h += ctx Wvo^T
      if (on(3)) {
        const bool two_stage = W.wq1 != nullptr && !xattn_merged_qt();
        // bytes: the weight(s) once + the residual rows in + the split-bf16 queries out; two-stage: both factors padded to 64 rows per head
        ProfScope p(this, "dec_crossq_gemm", two_stage ? 2.0 * 2.0 * M * D * 64.0 * Hh : 2.0 * M * D * D * Hh,
                    (two_stage ? 2.0 * Hh * 64.0 * D * 2.0 : w_dd * Hh) + M * D * 4.0 * (1 + Hh));
        if (two_stage) dec_crossq2(dH, W.wq1, W.wk2, M, Hh, D, reinterpret_cast<bf16_t*>(dq), s);
        else dec_gemm_ln_qt(dH, W.wqk, M, Hh, D, reinterpret_cast<bf16_t*>(dq), s);
      }
      if (on(4)) {
        // both products on all 16 MFMA columns (high / low halves of 8 heads); bytes: E once, qt in, ctx out
        ProfScope p(this, "dec_cross_attention", 2.0 * 2.0 * sT * D * 16, sT * D * 2.0 + M * D * Hh * 6.0);
        dec_cross_absorbed(reinterpret_cast<const bf16_t*>(dq), ENC_.as<bf16_t>(), clips, M, D, Hh, dao, s, xattn_stream_nt(shared_gpu_));
      }
      if (on(5)) {
        ProfScope p(this, "dec_ctx_resid_gemm", 2.0 * M * D * D * Hh, w_dd * Hh + M * D * (2.0 * Hh + 8.0));
        dec_gemm_resid(dao, W.wvo, nullptr, M, D, Hh * D, dH, s);
      }
    } else if (g.split_cross) {
      // single-clip latency path (k_dec_small.hip): one wave per (64-key slice, head, clip) with LayerNorm + query projection
      // inside, then the output projection with the merge of the slices as its prologue
      if (on(4)) {
        ProfScope p(this, "dec_cross_attention", 4.0 * sT * D + 2.0 * M * D * D, sT * D * 2.0 * kv_bytes() + w_dd + M * D * 4.0);
        dec_cross_split(dH, W.wq_c_rm, KTl, VTl, clips, M, D, Hh, g.xs_max, g.xpart.as<float>(), s);
      }
      if (on(5)) {
        ProfScope p(this, "dec_proj_resid_gemm", 2.0 * M * D * D, w_dd + M * D * 10.0);
        dec_merge_resid(g.xpart.as<float>(), W.wo_c, M, D, Hh, g.xs_max, dH, s);
      }
    } else if (g.loop_cross) {
      // the same arithmetic, one workgroup per (clip, head) looping over the slices; the standard projection follows below
      if (on(4)) {
        ProfScope p(this, "dec_cross_attention", 4.0 * sT * D + 2.0 * M * D * D, sT * D * 2.0 * kv_bytes() + w_dd + M * D * 4.0);
        dec_cross_looped(dH, W.wq_c_rm, KTl, VTl, clips, M, D, Hh, dao, s);
      }
    } else if (fuse_q && D <= 512 && M < 64 && !capture_cross_ && !uniform_kernels_) {  // latency-bound regime only (see k_attn.hip)
      // LayerNorm + query projection of the clip's row run inside the attention kernel
      if (on(4)) {
        ProfScope p(this, "dec_cross_attention", 4.0 * sT * D + 2.0 * M * D * D, sT * D * 2.0 * kv_bytes() + w_dd + M * D * 4.0);
        dec_cross_attention_fused_q(dH, W.wq_c_rm, KTl, VTl, clips, M, D, Hh, dao, s, kdq(l), vdq(l));
      }
    } else {
      if (on(3)) {
        ProfScope p(this, "dec_crossq_gemm", 2.0 * M * D * D, w_dd + M * D * 8.0);
        dec_gemm_ln_f32(dH, W.wq_c, M, D, D, dq, s);
      }
      if (capture_cross_ && on(4))
        dec_cross_attention_probs(dq, KTl, clips, pos, M, D, Hh, cfg_.dec_layers, l, cross_smax_, cross_tcap_,
                                  cross_probs_.as<float>() + (size_t)g.first * cfg_.dec_layers * Hh * cross_smax_ * cross_tcap_, s);
      if (on(4)) {
        ProfScope p(this, "dec_cross_attention", 4.0 * sT * D, sT * D * 2.0 * kv_bytes());
        dec_cross_attention(dq, KTl, VTl, clips, M, D, Hh, dao, s, kdq(l), vdq(l));
      }
    }
    if (on(5) && !absorbed_ && !g.split_cross) {
      ProfScope p(this, "dec_proj_resid_gemm", 2.0 * M * D * D, w_dd + M * D * 10.0);
      dec_gemm_resid(dao, W.wo_c, nullptr, M, D, D, dH, s);
    }
    if (on(6)) {
      ProfScope p(this, "dec_fc1_swiglu_gemm", 2.0 * M * D * 2 * F, 2.0 * 2 * F * D + M * (D * 4.0 + F * 2.0));
      dec_gemm_ln_swiglu(dH, W.fc1, W.b1, M, F, D, dz, s);
    }
    if (on(7)) {
      ProfScope p(this, "dec_fc2_resid_gemm", 2.0 * M * D * F, 2.0 * F * D + M * (F * 2.0 + D * 8.0));
      dec_gemm_resid(dz, W.fc2, W.b2, M, D, F, dH, s);
    }
  }
  if (M >= 128 || uniform_kernels_) {  // batch large enough for the LDS-tiled MFMA kernel: final LN once, then [M,D] x [V,D]^T
    if (on(8)) {
      ProfScope p(this, "dec_final_layernorm", 0, M * D * 6.0);
      dec_final_layernorm(dH, dec_ln_, M, D, g.dy.as<bf16_t>(), s);
    }
    if (!on(9)) {
    } else if (g.fused_argmax) {
      // nobody reads the logits: reduce every 128 x 208 tile to (max, first index) per row in the GEMM epilogue
      ProfScope p(this, "dec_lm_head_gemm", 2.0 * M * D * V, 2.0 * V * D + M * (double)gemm_argmax_tiles(V) * 8);
      gemm_argmax_partials(g.dy.as<bf16_t>(), D, embed_bf16_, M, V, D, g.pval.as<float>(), g.pidx.as<int>(), s);
    } else {
      ProfScope p(this, "dec_lm_head_gemm", 2.0 * M * D * V, 2.0 * V * D + M * (double)V * 4);
      gemm_logits_f32(g.dy.as<bf16_t>(), D, embed_bf16_, M, V, D, g.logits.as<float>(), s);
    }
  } else if (on(9)) {
    ProfScope p(this, "dec_lm_head_gemm", 2.0 * M * D * V, 2.0 * V * D + M * (double)V * 4);
    dec_gemm_logits(dH, embed_head_folded_, M, V, D, g.logits.as<float>(), s);
  }
}

// Per-launch cost of every decode kernel group INSIDE a dependent chain of a replayed hipGraph -- the way the decode
// loop runs them.  HIP-event scopes add ~4.8 us to every launch and rocprofv3 reports >= 4.3 us even for an empty kernel
// (its per-dispatch instrumentation), so neither can resolve kernels whose real marginal cost is 2-6 us.  For each group
// a graph of `reps` decode steps containing ONLY that group's launches (8 layers x its launches per layer, same
// arguments as the real step) is captured and replayed; ms / launches of the "chain_*" entries is the marginal cost.
// The decode state is garbage afterwards (residuals accumulate): encode + decode again before using results.
void Engine::profile_decode_chain(int reps) {
  MSH_HIP(hipSetDevice(device_));
  if (!encoded_ || groups_.empty() || groups_[0]->M != (int)n_clips_ || !groups_[0]->has_state)
    throw std::runtime_error("profile_decode_chain: encode and decode a batch first");
  if (prof_on_) throw std::runtime_error("profile_decode_chain: switch the event profiler off first");
  if (reps < 1) reps = 1;
  DecodeGroup& g = *groups_[0];
  const int V = cfg_.vocab, D = cfg_.hidden;
  static const char* names[] = {"dec_qkv_gemm", "dec_self_attention", "dec_proj_resid_gemm", "dec_crossq_gemm",
                                "dec_cross_attention", "dec_proj_resid_gemm#cross", "dec_fc1_swiglu_gemm", "dec_fc2_resid_gemm",
                                "dec_final_layernorm", "dec_lm_head_gemm", "dec_argmax_advance", "empty_step"};
  MSH_HIP(hipStreamSynchronize(g.stream));
  hipEvent_t a = get_event(), b = get_event();
  for (int id = 0; id < 12; ++id) {
    int launches = 0;
    hipGraph_t gr = nullptr;
    hipGraphExec_t ge = nullptr;
    {
      std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
      MSH_HIP(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
      for (int r = 0; r < reps; ++r) {
        if (id < 10) {
          step_only_ = id;
          decode_step_enqueue(g);
          step_only_ = -1;
        } else if (id == 10) {
          if (g.fused_argmax)
            decode_advance_partials(g.pval.as<float>(), g.pidx.as<int>(), g.argmax_tiles, g.M,
                                    clips_d_.as<ClipMeta>() + g.first, g.state, embed_f32_, D, g.dH.as<float>(), g.stream);
          else
            decode_advance(g.logits.as<float>(), g.M, V, clips_d_.as<ClipMeta>() + g.first, g.state, embed_f32_, D,
                           g.dH.as<float>(), g.stream);
        }
      }
      MSH_HIP(hipStreamEndCapture(g.stream, &gr));
      size_t n_nodes = 0;
      MSH_HIP(hipGraphGetNodes(gr, nullptr, &n_nodes));
      launches = (int)n_nodes;
      MSH_HIP(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      MSH_HIP(hipGraphDestroy(gr));
    }
    if (launches > 0) {
      MSH_HIP(hipGraphLaunch(ge, g.stream));  // warm
      MSH_HIP(hipEventRecord(a, g.stream));
      MSH_HIP(hipGraphLaunch(ge, g.stream));
      MSH_HIP(hipGraphLaunch(ge, g.stream));
      MSH_HIP(hipEventRecord(b, g.stream));
      MSH_HIP(hipStreamSynchronize(g.stream));
      float ms = 0.f;
      MSH_HIP(hipEventElapsedTime(&ms, a, b));
      // (absorbed cross-attention: group 5 is the wide-K context GEMM, a different kernel from the self-attention o-proj)
      const std::string name = std::string("chain_") + (id == 5 && absorbed_ ? "dec_ctx_resid_gemm" : names[id]);
      auto it = prof_idx_.find(name);
      if (it == prof_idx_.end()) {
        prof_idx_[name] = (int)prof_.size();
        prof_.push_back(ProfEntry{name, 0, 0, 0, 0});
        it = prof_idx_.find(name);
      }
      prof_[it->second].ms += ms;
      prof_[it->second].launches += 2ull * launches;
    }
    MSH_HIP(hipGraphExecDestroy(ge));
  }
  // Developer: MSH_CHAIN_MASKS = comma-separated bit masks of kernel groups (bit i = group i of `names`, per layer): each
  // mask is timed as its own replayed chain ("chainmask_<mask>", ms per decode STEP in `ms / launches * groups-in-mask * 8`
  // terms: launches counts the graph's nodes) -- what a SEQUENCE of different kernels costs, against the sum of its members.
  if (const char* masks = dev_getenv("MSH_CHAIN_MASKS")) {
    std::string list(masks);
    size_t pos = 0;
    while (pos < list.size()) {
      const size_t comma = list.find(',', pos);
      const std::string tok = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
      pos = comma == std::string::npos ? list.size() : comma + 1;
      const unsigned mask = (unsigned)strtoul(tok.c_str(), nullptr, 0) & 0xffu;
      if (mask == 0) continue;
      hipGraph_t gr = nullptr;
      hipGraphExec_t ge = nullptr;
      size_t n_nodes = 0;
      {
        std::lock_guard<std::mutex> structure_lock(device_structure_mutex());
        MSH_HIP(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
        step_mask_ = mask;
        try {
          for (int r = 0; r < reps; ++r) decode_step_enqueue(g);
        } catch (...) {   // never leave the stream in capture mode
          step_mask_ = 0;
          (void)hipStreamEndCapture(g.stream, &gr);
          if (gr != nullptr) (void)hipGraphDestroy(gr);
          throw;
        }
        step_mask_ = 0;
        MSH_HIP(hipStreamEndCapture(g.stream, &gr));
        MSH_HIP(hipGraphGetNodes(gr, nullptr, &n_nodes));
        const hipError_t inst = hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        MSH_HIP(inst);
      }
      if (n_nodes > 0) {
        MSH_HIP(hipGraphLaunch(ge, g.stream));
        MSH_HIP(hipEventRecord(a, g.stream));
        MSH_HIP(hipGraphLaunch(ge, g.stream));
        MSH_HIP(hipGraphLaunch(ge, g.stream));
        MSH_HIP(hipEventRecord(b, g.stream));
        MSH_HIP(hipStreamSynchronize(g.stream));
        float ms = 0.f;
        MSH_HIP(hipEventElapsedTime(&ms, a, b));
        const std::string name = "chainmask_" + tok;
        if (prof_idx_.find(name) == prof_idx_.end()) {
          prof_idx_[name] = (int)prof_.size();
          prof_.push_back(ProfEntry{name, 0, 0, 0, 0});
        }
        ProfEntry& pe = prof_[prof_idx_[name]];
        pe.ms += ms;
        pe.launches += 2ull * n_nodes;
      }
      MSH_HIP(hipGraphExecDestroy(ge));
    }
  }
  event_pool_.push_back(a);
  event_pool_.push_back(b);
}

int Engine::decode(int forced_steps, const int32_t* teacher, int teacher_stride, float* logits_out,
                   int max_logit_steps, int32_t* tokens_out, int32_t* counts_out, int tokens_stride) {
  MSH_HIP(hipSetDevice(device_));
  if (!encoded_) throw std::runtime_error("decode() called before encode()");
  const int Mtot = (int)n_clips_, D = cfg_.hidden, F = cfg_.ffn, V = cfg_.vocab, Hh = cfg_.heads, dh = cfg_.head_dim();
  const bool forced = forced_steps >= 0;
  const int steps = forced ? forced_steps : max_steps_;
  if (steps < 1) throw std::invalid_argument("decode: need at least one step");
  if (steps > 504) throw std::invalid_argument("decode: step budget above 504 tokens is not supported");
  const int stride = steps + 1;
  if (tokens_out != nullptr && tokens_stride < stride)
    throw std::invalid_argument("decode: tokens_stride " + std::to_string(tokens_stride) + " < steps+1 = " +
                                std::to_string(stride));
  if (teacher != nullptr && teacher_stride < 1) throw std::invalid_argument("decode: bad teacher stride");
  // K / V cache rows per (clip, head) and the device-side token stride: capacity BUCKETS (a stride, not a length -- the
  // kernels read up to the current position), so that the captured graphs survive a changing step budget
  Smax_ = steps <= 72 ? 72 : steps <= 144 ? 144 : steps <= 288 ? 288 : 512;
  const int dstride = Smax_ + 8;   // >= steps + 1
  if (capture_cross_) {
    int tmax = 0;
    for (const ClipMeta& c : clips_h_) tmax = std::max(tmax, c.T);
    if (tmax > 2048) throw std::invalid_argument("word timestamps: clip longer than 2048 encoder frames");
    cross_tcap_ = round_up(tmax, 4);
    cross_smax_ = Smax_;
    cross_probs_.reserve((size_t)Mtot * cfg_.dec_layers * Hh * cross_smax_ * cross_tcap_ * sizeof(float));
    cross_counts_.clear();
  }

  // ---- groups ----
  const bool eager = prof_on_ || logits_out != nullptr || !use_graph_ || capture_cross_;
  int ngroups = 1;
  if (!eager) {
    ngroups = dec_groups_ > 0 ? dec_groups_ : 1;  // measured: extra streams only add per-kernel fixed cost
    if (ngroups > Mtot) ngroups = Mtot;
  }
  while ((int)groups_.size() < ngroups) {
    std::unique_ptr<DecodeGroup> g(new DecodeGroup());
    if (groups_.empty()) {
      g->stream = stream_;
    } else {
      MSH_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
      g->own_stream = true;
    }
    groups_.push_back(std::move(g));
  }
  if (enc_done_ == nullptr) MSH_HIP(hipEventCreateWithFlags(&enc_done_, hipEventDisableTiming));

  {  // clip metadata: the step budget is what decode_advance reads
    std::vector<ClipMeta> tmp = clips_h_;
    if (forced)
      for (ClipMeta& c : tmp) c.max_len = steps;  // fixed step count, EOS ignored (benchmark / parity mode)
    MSH_HIP(hipMemcpyAsync(clips_d_.p, tmp.data(), tmp.size() * sizeof(ClipMeta), hipMemcpyHostToDevice, stream_));
    MSH_HIP(hipStreamSynchronize(stream_));
  }
  MSH_HIP(hipEventRecord(enc_done_, stream_));

  std::vector<DecodeState> states(ngroups);
  for (int gi = 0; gi < ngroups; ++gi) {
    DecodeGroup& g = *groups_[gi];
    g.first = (int)((long)Mtot * gi / ngroups);
    g.M = (int)((long)Mtot * (gi + 1) / ngroups) - g.first;
    const int M = g.M;
    bool moved = false;
    const size_t M16 = (size_t)round_up(M, 16);  // the FM buffers hold whole 16-row MFMA tiles (kernels.h)
    moved |= g.dH.reserve(M16 * D * sizeof(float));
    const int xw = absorbed_ ? Hh : 1;   // absorbed cross-attention: the query and the context are `heads` rows of D per clip
    moved |= g.dq.reserve((size_t)M * D * xw * sizeof(float));
    moved |= g.dao.reserve(M16 * D * xw * sizeof(bf16_t));
    moved |= g.dz.reserve(M16 * F * sizeof(bf16_t));
    moved |= g.dy.reserve((size_t)M * D * sizeof(bf16_t));
    moved |= g.logits.reserve((size_t)M * V * sizeof(float));
    {
      const char* fe = dev_getenv("MSH_NO_FUSED_ARGMAX");   // (read per call: the tests switch it)
      const bool off = fe != nullptr && fe[0] == '1';
      // nobody reads the logits: the tiled LM head (from 128 clips on) reduces every 128 x 208 tile to (max, first index).
      // (The same on the split-K decode GEMM's 16 x 16 tiles was built and measured at one clip: head 6.5 against 6.4 us,
      // bookkeeping 4.85 against 4.25 -- that kernel's time is its chain of dependent accesses, not the scan; removed.)
      g.fused_argmax = !off && logits_out == nullptr && (M >= 128 || uniform_kernels_);
      g.argmax_tiles = gemm_argmax_tiles(V);
    }
    moved |= g.pval.reserve((size_t)M * g.argmax_tiles * sizeof(float));
    moved |= g.pidx.reserve((size_t)M * g.argmax_tiles * sizeof(int));
    {
      // single-clip latency path: split cross-attention (MSH_XSPLIT_M = largest batch that takes it, 0 = off)
      const char* xe = dev_getenv("MSH_XSPLIT_M");   // (read per call: the tests switch it)
      const int xsplit_m = xe != nullptr ? atoi(xe) : 4;
      const char* sfe = dev_getenv("MSH_SELF_FUSED_M");   // largest batch whose self-attention runs inside the o-proj launch (0 = off)
      const int self_m = sfe != nullptr ? atoi(sfe) : 1;   // (two clips: 7.7 us fused against 4.9 + 2.1 -- the waves take the clips in turn)
      g.self_fused = M <= std::min(self_m, 2) && dec_self_oproj_supported(D, Hh, M);
      int tmax = 1;
      for (int b = 0; b < M; ++b) tmax = std::max(tmax, (int)clips_h_[g.first + b].T);
      const int xs = dec_cross_split_slices(tmax);
      const bool small_ok = !uniform_kernels_ && !absorbed_ && !capture_cross_ && !kv_fp8_ && M < 64 && dec_cross_split_supported(D, Hh);
      g.split_cross = small_ok && M <= xsplit_m;
      const char* xl = dev_getenv("MSH_XLOOP");   // 0: batches above MSH_XSPLIT_M keep k_attn.hip's one-pass kernel (A/B measurements)
      g.loop_cross = small_ok && !g.split_cross && xs <= dec_cross_looped_max_slices() && !(xl != nullptr && xl[0] == '0');
      if (g.split_cross) {
        if (xs != g.xs_max) ++g.gen;   // baked into the captured launches
        g.xs_max = xs;
        moved |= g.xpart.reserve(dec_cross_split_part_floats(M, Hh, xs) * sizeof(float));
      }
    }
    moved |= g.cacheK.reserve((size_t)cfg_.dec_layers * M * Hh * Smax_ * dh * sizeof(bf16_t));
    moved |= g.cacheV.reserve((size_t)cfg_.dec_layers * M * Hh * Smax_ * dh * sizeof(bf16_t));
    moved |= g.tokens.reserve((size_t)M * dstride * sizeof(int32_t));
    moved |= g.counts.reserve((size_t)M * sizeof(int32_t));
    moved |= g.finished.reserve((size_t)M * sizeof(int32_t));
    moved |= g.scalars.reserve(16 * sizeof(int32_t));
    if (teacher) moved |= g.teacher.reserve((size_t)M * dstride * sizeof(int32_t));
    if (moved) ++g.gen;
    if (teacher) {
      std::vector<int32_t> t((size_t)M * dstride, 0);
      for (int b = 0; b < M; ++b)
        for (int i = 0; i < stride && i < teacher_stride; ++i)
          t[(size_t)b * dstride + i] = teacher[(size_t)(g.first + b) * teacher_stride + i];
      copy_blocking(g.teacher.p, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    DecodeState& st = states[gi];
    st.tokens = g.tokens.as<int32_t>();
    st.counts = g.counts.as<int32_t>();
    st.finished = g.finished.as<int32_t>();
    st.pos = g.scalars.as<int32_t>();
    st.n_active = g.scalars.as<int32_t>() + 1;
    st.forced = teacher ? g.teacher.as<int32_t>() : nullptr;
    st.stride = dstride;
    st.eos = cfg_.eos;
    st.ignore_eos = forced ? 1 : 0;
    const ClipMeta* clips = clips_d_.as<ClipMeta>() + g.first;

    if (g.own_stream) MSH_HIP(hipStreamWaitEvent(g.stream, enc_done_, 0));
    decode_begin(M, st, cfg_.bos, embed_f32_, D, g.dH.as<float>(), g.stream);
    if (!eager) {
      // everything baked into the captured kernel arguments (the step budget is not: it lives in the clips' metadata)
      const std::string key = std::to_string(M) + ":" + std::to_string(g.first) + ":" + std::to_string(Smax_) + ":" +
                              std::to_string(st.ignore_eos) + ":" + std::to_string(teacher != nullptr) + ":" +
                              std::to_string(absorbed_ ? 0 : kv_keys_) + ":" + std::to_string(g.fused_argmax) + ":" +
                              std::to_string(kv_fp8_) + ":" + std::to_string(absorbed_) + ":" + std::to_string(g.split_cross ? g.xs_max : 0) + ":" + std::to_string(g.self_fused) + ":" + std::to_string(g.loop_cross);
      if (g.graphs_ws_gen != ws_gen_ || g.graphs_gen != g.gen) {   // a workspace moved: every captured pointer is stale
        g.drop_graphs();
        g.graphs_ws_gen = ws_gen_, g.graphs_gen = g.gen;
      }
      // nothing step-dependent is a kernel argument (position, counters and ids live in device memory), so n consecutive
      // steps captured into ONE graph are n times the same nodes: the loop below replays that one wherever n whole steps
      // are left -- one graph launch (its start / end packets and the gap between two replays) per n steps instead of per step
      auto capture = [&](int n_steps, hipGraphExec_t* out) {
        hipGraph_t gr = nullptr;
        std::lock_guard<std::mutex> structure_lock(device_structure_mutex(device_));
        MSH_HIP(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
        try {
          for (int r = 0; r < n_steps; ++r) {
            decode_step_enqueue(g);
            if (g.fused_argmax)
              decode_advance_partials(g.pval.as<float>(), g.pidx.as<int>(), g.argmax_tiles, M, clips, st, embed_f32_, D,
                                      g.dH.as<float>(), g.stream);
            else
              decode_advance(g.logits.as<float>(), M, V, clips, st, embed_f32_, D, g.dH.as<float>(), g.stream);
          }
        } catch (...) {   // never leave the stream in capture mode: end it, drop the partial graph, report the real error
          (void)hipStreamEndCapture(g.stream, &gr);
          if (gr != nullptr) (void)hipGraphDestroy(gr);
          throw;
        }
        MSH_HIP(hipStreamEndCapture(g.stream, &gr));
        const hipError_t inst = hipGraphInstantiate(out, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        if (inst != hipSuccess) {
          *out = nullptr;
          MSH_HIP(inst);
        }
        ++g.captures;
      };
      DecodeGroup::GraphEntry* ent = nullptr;
      for (DecodeGroup::GraphEntry& ge : g.graphs)
        if (ge.key == key) ent = &ge;
      const bool first_shape = g.graphs.empty();
      if (ent == nullptr) {
        if (g.graphs.size() >= DecodeGroup::kGraphCache) {   // evict the least recently used shape
          size_t lru = 0;
          for (size_t k = 1; k < g.graphs.size(); ++k)
            if (g.graphs[k].last < g.graphs[lru].last) lru = k;
          if (g.graphs[lru].g1) MSH_HIP(hipGraphExecDestroy(g.graphs[lru].g1));
          if (g.graphs[lru].gn) MSH_HIP(hipGraphExecDestroy(g.graphs[lru].gn));
          g.graphs.erase(g.graphs.begin() + (long)lru);
        }
        g.graphs.emplace_back();
        ent = &g.graphs.back();
        ent->key = key;
        capture(1, &ent->g1);
      }
      ++ent->uses;
      ent->last = ++g.graph_clock;
      if (ent->gn == nullptr && graph_steps() > 1 && steps >= graph_steps() && (first_shape || ent->uses >= 2))
        capture(graph_steps(), &ent->gn);
      g.graph = ent->g1;
      g.graph_n = ent->gn;
    }
    g.n_active_h = M;
    g.state = st;
    g.has_state = true;
  }

  int steps_run = 0;
  bool have_n = true;   // every group holds the multi-step graph of this shape
  for (int gi = 0; gi < ngroups; ++gi) have_n = have_n && groups_[gi]->graph_n != nullptr;
  const int gsteps = graph_steps();   // 8 = the interval at which the host looks at the active-clip counter anyway
  for (int i = 0; i < steps; ++i) {
    // a whole block of steps in one replay: from a block boundary, when that many steps are left (the reference loop stops
    // at EOS / budget per clip -- finished clips are masked on the device, the host only checks between blocks)
    const bool block = !eager && have_n && gsteps > 1 && (i % gsteps) == 0 && i + gsteps <= steps;
    for (int gi = 0; gi < ngroups; ++gi) {
      DecodeGroup& g = *groups_[gi];
      if (g.n_active_h <= 0) continue;
      if (block && g.graph_n != nullptr) {
        MSH_HIP(hipGraphLaunch(g.graph_n, g.stream));
      } else if (!eager) {
        MSH_HIP(hipGraphLaunch(g.graph, g.stream));
      } else {
        decode_step_enqueue(g);
        if (logits_out != nullptr && i < max_logit_steps)
          MSH_HIP(hipMemcpyAsync(logits_out + (size_t)i * Mtot * V, g.logits.p, (size_t)Mtot * V * sizeof(float),
                                 hipMemcpyDeviceToHost, g.stream));
        if (g.fused_argmax) {
          ProfScope p(this, "dec_argmax_advance", 0, (double)g.M * g.argmax_tiles * 8);
          decode_advance_partials(g.pval.as<float>(), g.pidx.as<int>(), g.argmax_tiles, g.M,
                                  clips_d_.as<ClipMeta>() + g.first, states[gi], embed_f32_, D, g.dH.as<float>(), g.stream);
        } else {
          ProfScope p(this, "dec_argmax_advance", 0, (double)g.M * V * 4);
          decode_advance(g.logits.as<float>(), g.M, V, clips_d_.as<ClipMeta>() + g.first, states[gi], embed_f32_, D,
                         g.dH.as<float>(), g.stream);
        }
      }
    }
    if (block) {
      steps_run += gsteps;
      i += gsteps - 1;
    } else {
      ++steps_run;
    }
    if (!forced && ((i & 7) == 7) && i + 1 < steps) {
      int alive = 0;
      for (int gi = 0; gi < ngroups; ++gi) {
        DecodeGroup& g = *groups_[gi];
        if (g.n_active_h <= 0) continue;
        MSH_HIP(hipMemcpyAsync(&g.n_active_h, states[gi].n_active, sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
      }
      for (int gi = 0; gi < ngroups; ++gi) {
        MSH_HIP(hipStreamSynchronize(groups_[gi]->stream));
        alive += groups_[gi]->n_active_h > 0 ? 1 : 0;
      }
      if (alive == 0) break;
    }
  }
  for (int gi = 0; gi < ngroups; ++gi) {
    DecodeGroup& g = *groups_[gi];
    if (tokens_out != nullptr || counts_out != nullptr) {
      std::vector<int32_t> tk((size_t)g.M * dstride), cn(g.M);
      MSH_HIP(hipMemcpyAsync(tk.data(), g.tokens.p, tk.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
      MSH_HIP(hipMemcpyAsync(cn.data(), g.counts.p, cn.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
      MSH_HIP(hipStreamSynchronize(g.stream));
      for (int b = 0; b < g.M; ++b) {
        const int gb = g.first + b;
        if (counts_out) counts_out[gb] = cn[b];
        if (tokens_out)
          for (int i = 0; i < tokens_stride; ++i)
            tokens_out[(size_t)gb * tokens_stride + i] = i < cn[b] ? tk[(size_t)b * dstride + i] : -1;
      }
    } else {
      MSH_HIP(hipStreamSynchronize(g.stream));
    }
  }
  if (capture_cross_) {
    cross_counts_.resize(Mtot);
    for (int gi = 0; gi < ngroups; ++gi) {
      DecodeGroup& g = *groups_[gi];
      MSH_HIP(hipMemcpyAsync(cross_counts_.data() + g.first, g.counts.p, (size_t)g.M * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
      MSH_HIP(hipStreamSynchronize(g.stream));
    }
  }
  prof_flush();
  return steps_run;
}

void Engine::get_cross_attention(uint32_t clip, float* out, size_t cap, int dims[3]) {
  MSH_HIP(hipSetDevice(device_));
  if (cross_counts_.empty() || clip >= cross_counts_.size())
    throw std::runtime_error("cross-attention not captured (set_capture_cross_attention before decode)");
  const int LH = cfg_.dec_layers * cfg_.heads, steps = cross_counts_[clip] - 1, T = clips_h_.at(clip).T;
  dims[0] = LH, dims[1] = steps, dims[2] = T;
  if (out == nullptr || steps <= 0) return;
  if (cap < (size_t)LH * steps * T) throw std::invalid_argument("get_cross_attention: output buffer too small");
  // device rows are [clip][layer*head][Smax][Tcap]: one strided 2-D copy per (layer, head)
  const float* src = cross_probs_.as<float>() + (size_t)clip * LH * cross_smax_ * cross_tcap_;
  for (int lh = 0; lh < LH; ++lh)
    MSH_HIP(hipMemcpy2DAsync(out + (size_t)lh * steps * T, (size_t)T * sizeof(float), src + (size_t)lh * cross_smax_ * cross_tcap_,
                             (size_t)cross_tcap_ * sizeof(float), (size_t)T * sizeof(float), (size_t)steps, hipMemcpyDeviceToHost,
                             stream_));
  MSH_HIP(hipStreamSynchronize(stream_));
}

}  // namespace msh
