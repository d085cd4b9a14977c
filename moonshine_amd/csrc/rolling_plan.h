// Which clips of a batch that arrives in pieces go to the GPU when (MoonshineModel::rolling_add, transcriber.h): pure host
// logic, no device, so that it can be tested without one (msh_host_rolling_plan, tests/test_rolling_plan.py).
//
// A sub-batch decodes until its LAST clip is done and a decode step costs about the same for 16 rows as for 1024, so the
// cost of a call is the sum over its sub-batches of their longest clip: clips of similar length belong together
// (MoonshineModel::run_shard sorts a whole batch, longest first, and cuts it into sub-batches of batch_clips x 10 s of
// audio, at most 4 x batch_clips clips).  Here the batch arrives in pieces -- the segments of every chunk of clips the
// batch call's VAD has finished -- and every piece brings clips of every length.  Submitting "the longest 2560 s waiting"
// after every piece measured WORSE than waiting for everything (every sub-batch then holds a 10 s clip: 60 % more decode
// steps than the sorted cut).  So before the last piece only two kinds of sub-batch go out:
//   * a FULL sub-batch of clips of nearly one length (the shortest within 10 % of the longest), wherever it sits in the
//     sorted pool: the sorted cut of the whole call would form it anyway (real batches have such classes: clips the
//     detector did not split are all as long as the caller's clips);
//   * the SHORT clips -- the shortest ones that hold short_frac of the first piece's audio, about the share of the call's
//     work the GPU can do while the segmentation runs (~55 of ~340 ms for 2048 clips; 0.02 / 0.15 / 0.25 / 0.4 measured
//     347 / 342 / 383 / 374 ms per call) -- in full sub-batches: few decode steps each, and whatever short clips there
//     are while nothing has been submitted yet, so that the GPU starts at once;
// everything else waits for the last piece and is cut sorted, longest first, like a whole batch.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace msh_host {

class RollingPlanner {
 public:
  explicit RollingPlanner(int batch_clips, double short_frac = 0.15, bool narrow_runs = true);
  // `count` more clips of n[i] samples (16 kHz); their indices continue from the previous call.  Returns the sub-batches
  // to submit now, in submission order, each a list of clip indices, longest clip first.  last: nothing may stay behind.
  std::vector<std::vector<uint32_t>> add(const uint64_t* n, size_t count, bool last);
  size_t waiting() const { return pool_.size(); }
  uint32_t clips_seen() const { return next_idx_; }
  uint64_t audio_cap() const { return audio_cap_; }
  uint32_t clip_cap() const { return clip_cap_; }

 private:
  struct Clip {
    uint32_t idx;
    uint64_t n;
  };
  uint32_t cut_at(size_t lo, uint64_t* sum) const;   // the cut run_shard makes at position lo of the sorted pool
  std::vector<uint32_t> take(size_t lo, uint32_t m);
  std::vector<Clip> pool_;   // waiting clips, longest first
  uint32_t bc_, clip_cap_, next_idx_ = 0;
  uint64_t audio_cap_, short_len_ = 0;
  double short_frac_;
  bool narrow_runs_, submitted_any_ = false;
};

}  // namespace msh_host
