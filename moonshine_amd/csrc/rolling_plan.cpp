#include "rolling_plan.h"

#include <algorithm>

namespace msh_host {

RollingPlanner::RollingPlanner(int batch_clips, double short_frac, bool narrow_runs, int lanes, double steps_per_second)
    : bc_((uint32_t)std::max(1, batch_clips)), short_frac_(short_frac), steps_(steps_per_second), lanes_(std::max(1, lanes)),
      narrow_runs_(narrow_runs) {
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

// The pool (sorted, longest first) as K consecutive runs of about equal cost_ms, K = the plain cut's count rounded up to a
// multiple of the lanes.  Greedy cut under a cost limit, the smallest limit that needs at most K runs found by bisection; the
// plain cut's caps (audio once a run holds batch_clips clips, clips) still hold, so no run is larger than one of those.
std::vector<uint32_t> RollingPlanner::balanced_sizes() const {
  std::vector<uint32_t> plain;
  for (size_t lo = 0; lo < pool_.size();) {
    uint64_t sum = 0;
    const uint32_t m = cut_at(lo, &sum);
    plain.push_back(m);
    lo += m;
  }
  if (lanes_ <= 1 || plain.size() <= 1) return plain;
  const size_t K = (plain.size() + (size_t)lanes_ - 1) / (size_t)lanes_ * (size_t)lanes_;
  auto cut_under = [&](double limit, std::vector<uint32_t>* sizes) {
    sizes->clear();
    for (size_t lo = 0; lo < pool_.size();) {
      uint32_t m = 0;
      uint64_t sum = 0;
      while (lo + m < pool_.size() && m < clip_cap_) {
        const uint64_t n = pool_[lo + m].n;
        if (m >= bc_ && sum + n > audio_cap_) break;
        if (m >= 1 && cost_ms(sum + n, pool_[lo].n) > limit) break;
        sum += n;
        ++m;
      }
      sizes->push_back(m);
      lo += m;
    }
  };
  double total = 0.0, lo_c = 0.0;
  {
    uint64_t audio = 0;
    for (const Clip& c : pool_) audio += c.n, lo_c = std::max(lo_c, cost_ms(c.n, c.n));
    total = cost_ms(audio, pool_.front().n);
  }
  const double hi0 = std::max(total, lo_c), lo0 = lo_c;
  auto cut_into = [&](size_t k, std::vector<uint32_t>* sizes) {   // the smallest cost limit that needs at most k runs
    double lo_b = lo0, hi_b = hi0;
    std::vector<uint32_t> trial;
    cut_under(hi_b, sizes);
    if (sizes->size() > k) return false;
    for (int it = 0; it < 40; ++it) {
      const double mid = 0.5 * (lo_b + hi_b);
      cut_under(mid, &trial);
      if (trial.size() <= k) hi_b = mid, *sizes = trial;
      else lo_b = mid;
    }
    return true;
  };
  auto spread = [&](const std::vector<uint32_t>& sizes) {   // costliest run / mean run
    double mx = 0.0, sum = 0.0;
    size_t lo = 0;
    for (uint32_t m : sizes) {
      uint64_t audio = 0;
      for (uint32_t i = 0; i < m; ++i) audio += pool_[lo + i].n;
      const double c = cost_ms(audio, pool_[lo].n);
      mx = std::max(mx, c), sum += c;
      lo += m;
    }
    return mx * (double)sizes.size() / sum;
  };
  // More runs cost decode steps (every run decodes until its longest clip is done), fewer leave lanes idle at the end: take
  // the smallest multiple of the lanes whose costliest run is within 25 % of the mean, at most twice the plain count.
  std::vector<uint32_t> best, cand;
  for (size_t k = K; k <= 2 * plain.size() + (size_t)lanes_; k += (size_t)lanes_) {
    if (!cut_into(k, &cand)) continue;
    if (best.empty()) best = cand;
    if (spread(cand) <= 1.25) return cand;
    if (spread(cand) < spread(best)) best = cand;
    if (!extra_runs_) break;
  }
  return best.empty() ? plain : best;
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
  if (last && lanes_ > 1) {   // the balanced last cut (rolling_plan.h)
    for (uint32_t m : balanced_sizes()) out.push_back(take(0, m));
    return out;
  }
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
