// Batches in flight: N engines ("lanes") on one GPU, each with its own HIP stream, workspace and host thread, sharing
// one copy of the weights.  A transcription call is an encoder pass (large MFMA kernels that fill the chip) followed by
// ~65 decode steps of short, latency-bound kernels that leave most CUs idle; with two or three independent batches in
// flight the encoder of one batch runs inside the gaps of the other's decode loop (measured at 256 x 10 s clips:
// 46.0k -> 59.0k -> 63.1k audio-s/s for 1 / 2 / 3 lanes, tokens identical; profiles/r03a_*.txt).
// The reference has no counterpart: its batch dimension is 1 and calls are serialised by processing_mutex
// (core/moonshine-model.cpp:229).
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace msh {

class BatchPipeline {
 public:
  // `primary` must be loaded and outlive the pipeline; the lanes share its weight buffers.
  BatchPipeline(Engine& primary, int device, int lanes);
  ~BatchPipeline();
  int lanes() const { return (int)lanes_.size(); }

  // Queue one batch (encode + greedy decode, the arguments of Engine::encode / Engine::decode).  The clip memory and the
  // output arrays must stay valid until wait() returns; the pointer / length arrays are copied here.
  int64_t submit(const float* const* pcm, const uint64_t* n_samples, uint32_t count, bool on_device, float mtps,
                 int forced_steps, int32_t* tokens_out, int32_t* counts_out, int tokens_stride);
  // Blocks until the batch is done; rethrows what the lane threw.  A ticket can be waited for once.
  void wait(int64_t ticket);

 private:
  struct Job {
    int64_t ticket = 0;
    std::vector<const float*> pcm;
    std::vector<uint64_t> n_samples;
    bool on_device = false;
    float mtps = 6.5f;
    int forced_steps = -1;
    int32_t *tokens_out = nullptr, *counts_out = nullptr;
    int tokens_stride = 0;
    std::chrono::steady_clock::time_point submitted;
    bool done = false;
    std::exception_ptr error;
  };
  void worker(int lane);

  int device_;
  std::vector<std::unique_ptr<Engine>> lanes_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::deque<std::shared_ptr<Job>> queue_;
  std::map<int64_t, std::shared_ptr<Job>> jobs_;
  int64_t next_ticket_ = 0;
  bool stop_ = false;
};

}  // namespace msh
