#include "rolling_plan.h"

#include <algorithm>

namespace msh_host {

RollingPlanner::RollingPlanner(int batch_clips, double short_frac, bool narrow_runs)
    : bc_((uint32_t)std::max(1, batch_clips)), short_frac_(short_frac), narrow_runs_(narrow_runs) {
  clip_cap_ = (uint32_t)std::min<long>(4L * bc_, 1024);
  audio_cap_ = (uint64_t)bc_ * 160000ull;
}

uint32_t RollingPlanner::cut_at(size_t lo, uint64_t* sum) const {
  uint32_t m = 0;
  *sum = 0;
  while (lo + m < pool_.size() && m < clip_cap_) {
    // the first batch_clips clips always go in; further ones while the audio budget lasts
    if (m >= bc_ && *sum + pool_[lo + m].n > audio_cap_) break;
    *sum += pool_[lo + m].n;
    ++m;
  }
  return m;
}

std::vector<uint32_t> RollingPlanner::take(size_t lo, uint32_t m) {
  std::vector<uint32_t> ids(m);
  for (uint32_t i = 0; i < m; ++i) ids[i] = pool_[lo + i].idx;
  pool_.erase(pool_.begin() + (long)lo, pool_.begin() + (long)(lo + m));
  submitted_any_ = true;
  return ids;
}

std::vector<std::vector<uint32_t>> RollingPlanner::add(const uint64_t* n, size_t count, bool last) {
  std::vector<std::vector<uint32_t>> out;
  for (size_t i = 0; i < count; ++i) pool_.push_back({next_idx_++, n[i]});
  // (stable: clips of one length keep the order they came in, so the plan is a function of the lengths and pieces alone)
  std::stable_sort(pool_.begin(), pool_.end(), [](const Clip& a, const Clip& b) { return a.n > b.n; });
  if (short_len_ == 0 && !pool_.empty()) {
    uint64_t total = 0, acc = 0;
    for (const Clip& c : pool_) total += c.n;
    short_len_ = pool_.back().n;
    for (size_t k = pool_.size(); k-- > 0;) {   // ascending
      if ((double)(acc + pool_[k].n) > short_frac_ * (double)total) break;
      acc += pool_[k].n;
      short_len_ = pool_[k].n;
    }
  }
  if (!last && narrow_runs_) {
    for (size_t i = 0; i < pool_.size();) {
      uint64_t sum = 0;
      const uint32_t m = cut_at(i, &sum);
      if (sum * 10 >= audio_cap_ * 9 && pool_[i + m - 1].n * 10 >= pool_[i].n * 9)
        out.push_back(take(i, m));
      else
        ++i;
    }
  }
  size_t lo = 0;   // first candidate: everything on the last piece, else the first short clip
  if (!last)
    while (lo < pool_.size() && pool_[lo].n > short_len_) ++lo;
  uint64_t waiting = 0;
  for (size_t k = lo; k < pool_.size(); ++k) waiting += pool_[k].n;
  while (lo < pool_.size()) {
    // full sub-batches; everything on the last piece; and whatever short clips there are while the GPU has nothing yet
    if (!last && waiting < audio_cap_ && submitted_any_) break;
    uint64_t sum = 0;
    const uint32_t m = cut_at(lo, &sum);
    out.push_back(take(lo, m));
    waiting -= sum;
  }
  return out;
}

}  // namespace msh_host
