#include "silero_device.h"

#include <string.h>

#include <algorithm>
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
constexpr long kMaxHopsPerChunk = 65536;
}  // namespace

SileroDevice::SileroDevice(int device, const msh_host::SileroWeights& w) : device_(device) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) throw HipError("no HIP device available for the device VAD");
  if (device < 0 || device >= n_dev) throw HipError("invalid device index " + std::to_string(device));
  MSH_HIP(hipSetDevice(device_));
  MSH_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  try {
    upload_weights(w);
  } catch (...) {   // a constructor that throws runs no destructor: hand back what was taken so far
    {
      std::lock_guard<std::mutex> lock(device_structure_mutex());
      for (void* p : weights_) device_free(p);
    }
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
  DevBuf* bufs[] = {&audio_, &hop_base_, &clip_hop0_, &frames_, &stft_, &act_[0], &act_[1], &cols_, &gin_, &probs_};
  for (DevBuf* b : bufs) b->release();
  if (pinned_) (void)hipHostFree(pinned_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void SileroDevice::probabilities(const float* const* pcm, const uint64_t* n, size_t count, std::vector<std::vector<float>>* probs) {
  probs->assign(count, std::vector<float>());
  size_t c0 = 0;
  while (c0 < count) {
    long hops = 0;
    size_t c1 = c0;
    while (c1 < count) {
      const long h = (long)(n[c1] / kHop);
      if (c1 > c0 && hops + h > kMaxHopsPerChunk) break;
      hops += h;
      ++c1;
    }
    run_chunk(pcm, n, c0, c1, probs);
    c0 = c1;
  }
}

void SileroDevice::run_chunk(const float* const* pcm, const uint64_t* n, size_t c0, size_t c1,
                             std::vector<std::vector<float>>* probs) {
  MSH_HIP(hipSetDevice(device_));
  const size_t nc = c1 - c0;
  // flat audio: per clip 64 zeros of context, then its whole hops
  std::vector<long> clip_off(nc), clip_hop0(nc + 1, 0);
  long samples = 0, hops = 0;
  for (size_t i = 0; i < nc; ++i) {
    const long h = (long)(n[c0 + i] / kHop);
    if (h > 0 && pcm[c0 + i] == nullptr) throw std::invalid_argument("null audio pointer");
    clip_off[i] = samples;
    clip_hop0[i] = hops;
    samples += kContext + h * kHop;
    hops += h;
  }
  clip_hop0[nc] = hops;
  if (hops == 0) return;
  std::vector<long> hop_base((size_t)hops);
  for (size_t i = 0; i < nc; ++i)
    for (long j = clip_hop0[i]; j < clip_hop0[i + 1]; ++j) hop_base[(size_t)j] = clip_off[i] + (j - clip_hop0[i]) * kHop;
  // gather into pinned memory on a few host threads, one DMA
  const size_t bytes = (size_t)samples * sizeof(float);
  if (bytes > pinned_cap_) {
    if (pinned_) MSH_HIP(hipHostFree(pinned_));
    pinned_ = nullptr;
    pinned_cap_ = 0;
    MSH_HIP(hipHostMalloc(&pinned_, bytes + bytes / 8, hipHostMallocDefault));
    pinned_cap_ = bytes + bytes / 8;
  }
  float* stage = static_cast<float*>(pinned_);
  msh_host::parallel_for(nc, [&](size_t i) {
    float* dst = stage + clip_off[i];
    memset(dst, 0, kContext * sizeof(float));
    const size_t cnt = (size_t)(clip_hop0[i + 1] - clip_hop0[i]) * kHop;
    if (cnt > 0) memcpy(dst + kContext, pcm[c0 + i], cnt * sizeof(float));
  }, std::min(8u, msh_host::effective_cpus()));
  audio_.reserve(bytes + 4096);   // (the last hop's reflect padding reads inside its own 576 samples: no over-read)
  hop_base_.reserve((size_t)hops * sizeof(long));
  clip_hop0_.reserve((nc + 1) * sizeof(long));
  frames_.reserve((size_t)hops * 4 * 256 * sizeof(float));
  stft_.reserve((size_t)hops * 4 * 258 * sizeof(float));
  act_[0].reserve((size_t)hops * 129 * 4 * sizeof(float));
  act_[1].reserve((size_t)hops * 128 * 4 * sizeof(float));
  cols_.reserve((size_t)hops * 4 * kpad_[0] * sizeof(float));
  gin_.reserve((size_t)hops * 512 * sizeof(float));
  probs_.reserve((size_t)hops * sizeof(float));
  MSH_HIP(hipMemcpyAsync(audio_.p, stage, bytes, hipMemcpyHostToDevice, stream_));
  MSH_HIP(hipMemcpyAsync(hop_base_.p, hop_base.data(), (size_t)hops * sizeof(long), hipMemcpyHostToDevice, stream_));
  MSH_HIP(hipMemcpyAsync(clip_hop0_.p, clip_hop0.data(), (nc + 1) * sizeof(long), hipMemcpyHostToDevice, stream_));
  silero_frames(audio_.as<float>(), hop_base_.as<long>(), hops, frames_.as<float>(), stream_);
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
  silero_lstm(gin_.as<float>(), w_hh_, out_w_, out_b_, clip_hop0_.as<long>(), (int)nc, probs_.as<float>(), stream_);
  std::vector<float> all((size_t)hops);
  MSH_HIP(hipMemcpyAsync(all.data(), probs_.p, (size_t)hops * sizeof(float), hipMemcpyDeviceToHost, stream_));
  MSH_HIP(hipStreamSynchronize(stream_));   // also: hop_base / clip_hop0 / the pinned buffer may be reused from here on
  for (size_t i = 0; i < nc; ++i)
    (*probs)[c0 + i].assign(all.begin() + clip_hop0[i], all.begin() + clip_hop0[i + 1]);
}

}  // namespace msh
