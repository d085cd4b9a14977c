#include "silero_device.h"

#include <string.h>

#include <algorithm>
#include <chrono>
#include <stdexcept>

#include "host_utils.h"
#include "silero_kernels.h"

namespace msh {
namespace {
constexpr int kHop = msh_host::SileroVad::kHop, kContext = msh_host::SileroVad::kContext;
const int kConvIn[4] = {129, 128, 64, 64}, kConvOut[4] = {128, 64, 64, 128}, kConvStride[4] = {1, 2, 2, 1};
// Workspace per hop: frames 4096 B + |STFT| 4128 + im2col columns 6400 + activations 2064 + 2048 + gate inputs 2048 + audio
// ~2.3 KB = ~23 KB (plus DevBuf's 12.5 % growth slack), so 64 Ki hops = ~1.5 GB per chunk, 209 clips of 10 s.  (200000
// hops, the first value, was 4.6 GB per chunk on top of the engines' workspaces: the out-of-memory fallback to the host
// network would have come far earlier than intended.)
}  // namespace

SileroDevice::SileroDevice(int device, const msh_host::SileroWeights& w) : device_(device) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) throw HipError("no HIP device available for the device VAD");
  if (device < 0 || device >= n_dev) throw HipError("invalid device index " + std::to_string(device));
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  for (Slot& sl : slots_) MSH_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  try {
    upload_weights(w);
  } catch (...) {   // a constructor that throws runs no destructor: hand back what was taken so far
    {
      std::lock_guard<std::mutex> lock(device_structure_mutex());
      for (void* p : weights_) device_free(p);
    }
    for (Slot& sl : slots_) (void)hipEventDestroy(sl.done);
    (void)hipStreamDestroy(stream_);
    throw;
  }
}

void SileroDevice::upload_weights(const msh_host::SileroWeights& w) {
  auto upload = [&](const std::vector<float>& src) {
    void* p = nullptr;
    {
      std::lock_guard<std::mutex> lock(device_structure_mutex());
      p = device_alloc(std::max<size_t>(src.size(), 4) * sizeof(float));
    }
    weights_.push_back(p);
    copy_blocking(p, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
    return reinterpret_cast<float*>(p);
  };
  if (w.stft.size() != (size_t)258 * 256) throw std::runtime_error("Silero VAD weights: unexpected STFT basis size");
  basis_ = upload(w.stft);
  for (int i = 0; i < 4; ++i) {
    const int k = kConvIn[i] * 3;
    kpad_[i] = (k + 15) & ~15;
    if (w.conv_w[i].size() != (size_t)kConvOut[i] * k) throw std::runtime_error("Silero VAD weights: unexpected conv size");
    std::vector<float> padded((size_t)kConvOut[i] * kpad_[i], 0.f);
    for (int o = 0; o < kConvOut[i]; ++o) memcpy(&padded[(size_t)o * kpad_[i]], &w.conv_w[i][(size_t)o * k], (size_t)k * sizeof(float));
    conv_w_[i] = upload(padded);
    conv_b_[i] = upload(w.conv_b[i]);
  }
  w_ih_ = upload(w.w_ih);
  w_hh_ = upload(w.w_hh);
  std::vector<float> bs(512);
  for (int i = 0; i < 512; ++i) bs[i] = w.b_ih[i] + w.b_hh[i];
  bias_sum_ = upload(bs);
  out_w_ = upload(w.out_w);
  out_b_ = w.out_b;
}

SileroDevice::~SileroDevice() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  {
    std::lock_guard<std::mutex> lock(device_structure_mutex());
    for (void* p : weights_) device_free(p);
  }
  DevBuf* bufs[] = {&frames_, &stft_, &act_[0], &act_[1], &cols_, &gin_, &probs_};
  for (DevBuf* b : bufs) b->release();
  for (auto& b : arena_) b->release();
  for (Slot& sl : slots_) {
    sl.audio.release(), sl.audio16.release(), sl.hop_base.release(), sl.clip_hop0_d.release();
    if (sl.pinned) (void)hipHostFree(sl.pinned);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (stream_) (void)hipStreamDestroy(stream_);
}

void SileroDevice::probabilities(const float* const* pcm, const uint64_t* n, size_t count, std::vector<std::vector<float>>* probs,
                                 std::vector<const float*>* resident) {
  probs->assign(count, std::vector<float>());
  if (resident != nullptr) resident->assign(count, nullptr);
  // chunks of at most kMaxHopsPerSubmit hops; chunk k + 1 is staged and uploaded while chunk k's network runs
  std::vector<std::pair<size_t, size_t>> chunks;
  for (size_t c0 = 0; c0 < count;) {
    long hops = 0;
    size_t c1 = c0;
    while (c1 < count) {
      const long h = (long)(n[c1] / kHop);
      if (c1 > c0 && hops + h > kMaxHopsPerSubmit) break;
      hops += h;
      ++c1;
    }
    chunks.push_back({c0, c1});
    c0 = c1;
  }
  std::vector<int64_t> tickets(chunks.size(), -1);
  auto collect_into = [&](size_t k) {
    const size_t c0 = chunks[k].first, nc = chunks[k].second - c0;
    std::vector<float> flat;
    std::vector<const float*> res;
    collect(tickets[k], &flat, resident != nullptr ? &res : nullptr);
    size_t off = 0;
    for (size_t i = 0; i < nc; ++i) {
      const size_t h = (size_t)(n[c0 + i] / kHop);
      (*probs)[c0 + i].assign(flat.begin() + (long)off, flat.begin() + (long)(off + h));
      off += h;
      if (resident != nullptr) (*resident)[c0 + i] = res[i];
    }
  };
  try {
    for (size_t k = 0; k < chunks.size(); ++k) {
      tickets[k] = submit(pcm + chunks[k].first, n + chunks[k].first, chunks[k].second - chunks[k].first, resident != nullptr);
      if (k > 0) collect_into(k - 1);
    }
    if (!chunks.empty()) collect_into(chunks.size() - 1);
  } catch (...) {
    abandon();
    throw;
  }
}

void SileroDevice::release_audio() {
  bool busy = false;
  for (const Slot& sl : slots_) busy |= sl.busy;
  if (busy || next_collect_ != next_ticket_) abandon();
  arena_used_ = 0, arena_live_bytes_ = 0;
  // Keep what a typical call needs warm (kArenaKeep bytes of buffers, reused by the next call) and give the rest back: one
  // 2048-clip call otherwise left up to kArenaBudget = 8 GiB allocated for the transcriber's lifetime beside the engines'
  // workspaces.  (Nothing of the arena is in flight here: every ticket was collected or abandoned above.)
  size_t kept = 0;
  size_t keep_n = 0;
  for (; keep_n < arena_.size(); ++keep_n) {
    if (kept + arena_[keep_n]->cap > kArenaKeep) break;
    kept += arena_[keep_n]->cap;
  }
  for (size_t i = keep_n; i < arena_.size(); ++i) arena_[i]->release();   // (takes the device's structure lock itself)
  arena_.resize(keep_n);
}

void SileroDevice::abandon() {
  (void)hipStreamSynchronize(stream_);
  (void)hipGetLastError();
  for (Slot& sl : slots_) sl.busy = false;
  next_collect_ = next_ticket_;
}

void SileroDevice::sync_streams() {
  MSH_HIP(hipStreamSynchronize(stream_));
}

int64_t SileroDevice::submit(const float* const* pcm, const uint64_t* n, size_t nc, bool keep_audio) {
  return submit_any(pcm, nullptr, n, nc, keep_audio);
}
int64_t SileroDevice::submit_pcm16(const int16_t* const* pcm16, const uint64_t* n, size_t nc, bool keep_audio) {
  return submit_any(nullptr, pcm16, n, nc, keep_audio);
}

// exactly one of pcm / pcm16 is given.  16-bit clips cross PCIe as they are (half the bytes) and become fp32 (x / 32768, exact)
// on the device, in the buffer the network -- and, kept, the engine -- reads.
int64_t SileroDevice::submit_any(const float* const* pcm, const int16_t* const* pcm16, const uint64_t* n, size_t nc, bool keep_audio) {
  MSH_HIP(hipSetDevice(device_));
  Slot& sl = slots_[next_ticket_ % kSlots];
  if (sl.busy) throw std::invalid_argument("device VAD: two submissions are outstanding, collect one first");
  // flat audio: per clip 64 zeros of context, then its whole hops
  sl.clip_off.assign(nc, 0);
  sl.clip_hop0.assign(nc + 1, 0);
  long samples = 0, hops = 0;
  for (size_t i = 0; i < nc; ++i) {
    const long h = (long)(n[i] / kHop);
    if (h > 0 && (pcm16 != nullptr ? (const void*)pcm16[i] : (const void*)pcm[i]) == nullptr) throw std::invalid_argument("null audio pointer");
    sl.clip_off[i] = samples;
    sl.clip_hop0[i] = hops;
    samples += kContext + h * kHop;
    hops += h;
  }
  sl.clip_hop0[nc] = hops;
  if (hops > kMaxHopsPerSubmit && nc > 1)
    throw std::invalid_argument("device VAD: more than " + std::to_string(kMaxHopsPerSubmit) + " hops in one submission");
  sl.nc = nc;
  sl.hops = hops;
  sl.abuf = nullptr;
  sl.ticket = next_ticket_;
  if (hops == 0) {   // nothing to compute: the ticket is collected without touching the GPU
    sl.busy = true;
    return next_ticket_++;
  }
  // One pinned block per slot: the audio, the hop table, the clips' first hops -- and the place the probabilities come back to
  // (a copy to or from pageable memory blocks the host until it is done, which would serialise the two slots)
  const size_t bytes = (size_t)samples * sizeof(float);                                   // on the device, fp32
  const size_t stage_bytes = pcm16 != nullptr ? (size_t)samples * sizeof(int16_t) : bytes;   // what crosses PCIe
  const size_t off_hops = (stage_bytes + 255) & ~(size_t)255, off_clips = off_hops + (size_t)hops * sizeof(long);
  const size_t off_probs = (off_clips + (nc + 1) * sizeof(long) + 255) & ~(size_t)255, pinned_need = off_probs + (size_t)hops * sizeof(float);
  if (pinned_need > sl.pinned_cap) {
    if (sl.pinned) MSH_HIP(hipHostFree(sl.pinned));
    sl.pinned = nullptr;
    sl.pinned_cap = 0;
    MSH_HIP(hipHostMalloc(&sl.pinned, pinned_need + pinned_need / 8, hipHostMallocDefault));
    sl.pinned_cap = pinned_need + pinned_need / 8;
  }
  static const bool timing = getenv("MSH_HOST_TIMING") != nullptr;
  static const unsigned gather_threads = [] {   // MSH_SILERO_GATHER_THREADS: host threads of the pageable -> pinned gather
    const char* e = dev_getenv("MSH_SILERO_GATHER_THREADS");
    return e != nullptr && atoi(e) > 0 ? (unsigned)atoi(e) : std::min(8u, msh_host::effective_cpus());
  }();
  const auto t0 = std::chrono::steady_clock::now();
  char* pin = static_cast<char*>(sl.pinned);
  float* stage = reinterpret_cast<float*>(pin);
  int16_t* stage16 = reinterpret_cast<int16_t*>(pin);
  msh_host::parallel_for(nc, [&](size_t i) {
    const size_t cnt = (size_t)(sl.clip_hop0[i + 1] - sl.clip_hop0[i]) * kHop;
    if (pcm16 != nullptr) {
      int16_t* dst = stage16 + sl.clip_off[i];
      memset(dst, 0, kContext * sizeof(int16_t));
      if (cnt > 0) memcpy(dst + kContext, pcm16[i], cnt * sizeof(int16_t));
    } else {
      float* dst = stage + sl.clip_off[i];
      memset(dst, 0, kContext * sizeof(float));
      if (cnt > 0) memcpy(dst + kContext, pcm[i], cnt * sizeof(float));
    }
  }, gather_threads);
  long* hop_base = reinterpret_cast<long*>(pin + off_hops);
  for (size_t i = 0; i < nc; ++i)
    for (long j = sl.clip_hop0[i]; j < sl.clip_hop0[i + 1]; ++j) hop_base[j] = sl.clip_off[i] + (j - sl.clip_hop0[i]) * kHop;
  memcpy(pin + off_clips, sl.clip_hop0.data(), (nc + 1) * sizeof(long));
  sl.probs_host = reinterpret_cast<float*>(pin + off_probs);
  sl.gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  sl.t_enqueued = std::chrono::steady_clock::now();
  // the chunk's audio: the slot's scratch buffer, or -- residency asked for and room left -- a buffer of its own that outlives
  // the call
  DevBuf* abuf = &sl.audio;
  if (keep_audio && arena_live_bytes_ + bytes <= kArenaBudget) {
    if (arena_used_ == arena_.size()) arena_.emplace_back(new DevBuf());
    DevBuf* kept = arena_[arena_used_].get();
    try {   // no device memory for a buffer of its own is "no residency for this chunk" (the engine uploads its segments from
            // the host), not a failure of the device VAD
      kept->reserve(bytes + 4096);
      abuf = kept;
      ++arena_used_;
      arena_live_bytes_ += bytes;
    } catch (const std::exception&) {
      (void)hipGetLastError();
    }
  }
  // buffers only this slot's work touches (it was collected: nothing of it is in flight)
  abuf->reserve(bytes + 4096);   // (the last hop's reflect padding reads inside its own 576 samples: no over-read)
  sl.hop_base.reserve((size_t)hops * sizeof(long));
  sl.clip_hop0_d.reserve((nc + 1) * sizeof(long));
  // the network's workspaces are shared by both slots (their kernels run in order on one stream): growing one means the
  // other slot's kernels must be done with it first
  struct Need {
    DevBuf* b;
    size_t bytes;
  } needs[] = {{&frames_, (size_t)hops * 4 * 256 * sizeof(float)}, {&stft_, (size_t)hops * 4 * 258 * sizeof(float)},
               {&act_[0], (size_t)hops * 129 * 4 * sizeof(float)}, {&act_[1], (size_t)hops * 128 * 4 * sizeof(float)},
               {&cols_, (size_t)hops * 4 * kpad_[0] * sizeof(float)}, {&gin_, (size_t)hops * 512 * sizeof(float)},
               {&probs_, (size_t)hops * sizeof(float)}};
  bool grow = false;
  for (const Need& nd : needs) grow |= nd.b->cap < nd.bytes;
  if (grow) sync_streams();
  for (const Need& nd : needs) nd.b->reserve(nd.bytes);
  // Upload, network and read-back in order on the ONE stream this object owns.  (A copy stream of its own for the upload --
  // chunk k + 1's DMA beside chunk k's network, also in four pieces on four streams -- measured no faster: the network is ~1 ms
  // of a chunk's ~6; and every extra stream competes with the engine's lanes for the process's hardware queues -- with 8
  // queues and 4 lanes the wave pipeline went from 381 to 450 ms per call when this object took five of them.)
  if (pcm16 != nullptr) {
    sl.audio16.reserve(stage_bytes + 64);
    MSH_HIP(hipMemcpyAsync(sl.audio16.p, stage16, stage_bytes, hipMemcpyHostToDevice, stream_));
    silero_pcm16_to_f32(sl.audio16.as<int16_t>(), abuf->as<float>(), samples, stream_);
  } else {
    MSH_HIP(hipMemcpyAsync(abuf->p, stage, bytes, hipMemcpyHostToDevice, stream_));
  }
  MSH_HIP(hipMemcpyAsync(sl.hop_base.p, hop_base, (size_t)hops * sizeof(long), hipMemcpyHostToDevice, stream_));
  MSH_HIP(hipMemcpyAsync(sl.clip_hop0_d.p, pin + off_clips, (nc + 1) * sizeof(long), hipMemcpyHostToDevice, stream_));
  silero_frames(abuf->as<float>(), sl.hop_base.as<long>(), hops, frames_.as<float>(), stream_);
  silero_stft_mag(frames_.as<float>(), basis_, hops, stft_.as<float>(), act_[0].as<float>(), stream_);
  // conv stack: [129][4] -> [128][4] -> [64][2] -> [64][1] -> [128][1]
  int tin = 4, src = 0;
  for (int i = 0; i < 4; ++i) {
    silero_conv_relu(act_[src].as<float>(), kConvIn[i], tin, kConvStride[i], conv_w_[i], kpad_[i], conv_b_[i], kConvOut[i], hops,
                     cols_.as<float>(), act_[src ^ 1].as<float>(), stream_);
    tin = (tin - 1) / kConvStride[i] + 1;
    src ^= 1;
  }
  silero_gate_inputs(act_[src].as<float>(), w_ih_, bias_sum_, hops, gin_.as<float>(), stream_);
  silero_lstm(gin_.as<float>(), w_hh_, out_w_, out_b_, sl.clip_hop0_d.as<long>(), (int)nc, probs_.as<float>(), stream_);
  MSH_HIP(hipMemcpyAsync(sl.probs_host, probs_.p, (size_t)hops * sizeof(float), hipMemcpyDeviceToHost, stream_));
  MSH_HIP(hipEventRecord(sl.done, stream_));
  sl.abuf = abuf;
  sl.kept = abuf != &sl.audio;
  sl.busy = true;
  if (timing) sl.bytes = stage_bytes;
  return next_ticket_++;
}

void SileroDevice::collect(int64_t ticket, std::vector<float>* probs, std::vector<const float*>* resident) {
  MSH_HIP(hipSetDevice(device_));
  if (ticket < 0) throw std::invalid_argument("device VAD: unknown ticket " + std::to_string(ticket));
  Slot& sl = slots_[ticket % kSlots];
  if (!sl.busy || sl.ticket != ticket) throw std::invalid_argument("device VAD: unknown ticket " + std::to_string(ticket));
  if (ticket != next_collect_) throw std::invalid_argument("device VAD: tickets are collected in the order they were given");
  probs->clear();
  if (resident != nullptr) resident->assign(sl.nc, nullptr);
  // the slot only counts as collected once its work is known to be done: if the wait throws, the slot stays busy (a following
  // submit must not free or reuse its pinned block and device buffers under an upload / read-back still in flight) and the
  // caller's failure path (abandon) drains the stream
  if (sl.hops != 0) MSH_HIP(hipEventSynchronize(sl.done));
  ++next_collect_;
  sl.busy = false;
  if (sl.hops == 0) return;
  static const bool timing = getenv("MSH_HOST_TIMING") != nullptr;
  if (timing)
    MSH_LOGF("device VAD: %zu clips, %.0f MB gathered into pinned memory in %.2f ms, upload + network + read-back done %.2f ms after "
             "they were enqueued", sl.nc, (double)sl.bytes / 1e6, sl.gather_ms,
             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sl.t_enqueued).count());
  probs->assign(sl.probs_host, sl.probs_host + sl.hops);
  if (resident != nullptr && sl.kept)
    for (size_t i = 0; i < sl.nc; ++i)
      if (sl.clip_hop0[i + 1] > sl.clip_hop0[i]) (*resident)[i] = sl.abuf->as<float>() + sl.clip_off[i] + kContext;
}

}  // namespace msh
