#include "pipeline.h"

#include "host_utils.h"

#include <algorithm>
#include <chrono>
#include <stdexcept>

namespace msh {

BatchPipeline::BatchPipeline(Engine& primary, int device, int lanes) : device_(device) {
  if (lanes < 1 || lanes > 8) throw std::invalid_argument("batches in flight must be 1..8");
  if (!primary.loaded()) throw std::runtime_error("batches in flight: load the weights first");
  for (int i = 0; i < lanes; ++i) {
    std::unique_ptr<Engine> e(new Engine(device));
    e->share_weights_from(primary);
    e->set_shared_gpu(lanes > 1);
    lanes_.push_back(std::move(e));
  }
  for (int i = 0; i < lanes; ++i) threads_.emplace_back([this, i] { worker(i); });
}

BatchPipeline::~BatchPipeline() {
  {
    // batches still queued are cancelled (their output arrays may be gone with the caller); running ones finish
    std::lock_guard<std::mutex> lock(mu_);
    stop_ = true;
    for (const std::shared_ptr<Job>& j : queue_) {
      j->error = std::make_exception_ptr(std::runtime_error("batch cancelled: the lanes were torn down"));
      j->done = true;
    }
    queue_.clear();
  }
  cv_work_.notify_all();
  cv_done_.notify_all();
  for (std::thread& t : threads_) t.join();
}

int64_t BatchPipeline::submit(const float* const* pcm, const uint64_t* n_samples, uint32_t count, bool on_device, float mtps,
                              int forced_steps, int32_t* tokens_out, int32_t* counts_out, int tokens_stride) {
  if (count == 0 || pcm == nullptr || n_samples == nullptr) throw std::invalid_argument("submit: empty batch");
  std::shared_ptr<Job> j(new Job());
  j->pcm.assign(pcm, pcm + count);
  j->n_samples.assign(n_samples, n_samples + count);
  j->on_device = on_device;
  j->mtps = mtps;
  j->forced_steps = forced_steps;
  j->tokens_out = tokens_out;
  j->counts_out = counts_out;
  j->tokens_stride = tokens_stride;
  j->submitted = std::chrono::steady_clock::now();
  {
    std::lock_guard<std::mutex> lock(mu_);
    j->ticket = next_ticket_++;
    jobs_[j->ticket] = j;
    queue_.push_back(j);
  }
  cv_work_.notify_one();
  return j->ticket;
}

void BatchPipeline::wait(int64_t ticket) {
  std::shared_ptr<Job> j;
  {
    std::unique_lock<std::mutex> lock(mu_);
    auto it = jobs_.find(ticket);
    if (it == jobs_.end()) throw std::invalid_argument("wait: unknown ticket " + std::to_string(ticket));
    j = it->second;
    cv_done_.wait(lock, [&] { return j->done; });
    jobs_.erase(ticket);
  }
  if (j->error) std::rethrow_exception(j->error);
}

void BatchPipeline::worker(int lane) {
  Engine& eng = *lanes_[lane];
  // more than one GPU in this process: the lane's host thread stays on its GPU's NUMA node (host_utils.h)
  int visible = 0;
  if (hipGetDeviceCount(&visible) == hipSuccess && visible > 1) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device_) == hipSuccess) (void)msh_host::pin_thread_to_gpu_node(bus);
    (void)hipGetLastError();
  }
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lock(mu_);
      cv_work_.wait(lock, [&] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) return;  // stop requested and nothing left to do
      j = queue_.front();
      queue_.pop_front();
    }
    try {
      static const bool timing = getenv("MSH_HOST_TIMING") != nullptr;   // one line per sub-batch, to the log
      const auto t0 = std::chrono::steady_clock::now();
      eng.encode(j->pcm.data(), j->n_samples.data(), (uint32_t)j->pcm.size(), j->on_device, j->mtps);
      const auto t1 = std::chrono::steady_clock::now();
      eng.decode(j->forced_steps, nullptr, 0, nullptr, 0, j->tokens_out, j->counts_out, j->tokens_stride);
      if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        uint64_t sum = 0, longest = 0, shortest = ~0ull;
        for (uint64_t n : j->n_samples) sum += n, longest = std::max(longest, n), shortest = std::min(shortest, n);
        const uint64_t captures = eng.debug_read("graph_captures", nullptr, 0);
        MSH_LOGF("lane %d: sub-batch %lld of %zu clips (%.0f s of audio, %.2f .. %.2f s, %s) encode %.1f ms, decode %.1f ms, started %.1f ms "
                 "after its submission; %llu decode graphs captured by this lane so far",
                 lane, (long long)j->ticket, j->pcm.size(), (double)sum / 16000.0, (double)shortest / 16000.0, (double)longest / 16000.0,
                 j->on_device ? "device audio" : "host audio", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(t2 - t1).count(),
                 std::chrono::duration<double, std::milli>(t0 - j->submitted).count(), (unsigned long long)captures);
      }
    } catch (...) {
      j->error = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lock(mu_);
      j->done = true;
    }
    cv_done_.notify_all();
  }
}

}  // namespace msh
