// Per-kernel-group timing with HIP events on one stream: the role of the reference's log_ort_run option
// (core/ort-utils/ort-utils.cpp:256-288), shared by the engines that do not carry their own.  A Scope brackets the
// launches of one group; entries accumulate the event time, the launch count and the group's ALGORITHMIC flops / bytes
// (what bench.py prices against the MFMA / HBM peaks).  An event scope itself costs ~4.8 us: figures for kernels of a few
// microseconds are upper bounds (see msh_profile_event_overhead_ms in include/moonshine_hip.h).
#pragma once

#include <map>
#include <string>
#include <vector>

#include "msh_common.h"

namespace msh {

struct ProfEntry {
  std::string name;
  double ms = 0;
  uint64_t launches = 0;
  double flops = 0;  // algorithmic flops over all launches
  double bytes = 0;  // algorithmic HBM bytes over all launches
};

class ScopeProfiler {
 public:
  ~ScopeProfiler() {
    for (auto& r : pending_) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    for (hipEvent_t e : pool_) (void)hipEventDestroy(e);
  }
  bool on() const { return on_; }
  void enable(bool v, hipStream_t s) {
    flush(s);
    on_ = v;
  }
  void reset(hipStream_t s) {
    flush(s);
    entries_.clear();
    index_.clear();
  }
  std::vector<ProfEntry> get(hipStream_t s) {
    flush(s);
    return entries_;
  }
  class Scope {
   public:
    Scope(ScopeProfiler* p, hipStream_t s, const char* name, double flops, double bytes) : p_(p), s_(s) {
      if (p_ == nullptr || !p_->on_) {
        p_ = nullptr;
        return;
      }
      auto it = p_->index_.find(name);
      if (it == p_->index_.end()) {
        idx_ = (int)p_->entries_.size();
        p_->index_[name] = idx_;
        ProfEntry e;
        e.name = name;
        p_->entries_.push_back(e);
      } else {
        idx_ = it->second;
      }
      ProfEntry& e = p_->entries_[idx_];
      e.flops += flops;
      e.bytes += bytes;
      e.launches += 1;
      a_ = p_->event();
      b_ = p_->event();
      MSH_HIP(hipEventRecord(a_, s_));
    }
    ~Scope() {
      if (p_ == nullptr) return;
      (void)hipEventRecord(b_, s_);
      p_->pending_.push_back({idx_, a_, b_});
    }
    Scope(const Scope&) = delete;
    Scope& operator=(const Scope&) = delete;

   private:
    ScopeProfiler* p_;
    hipStream_t s_;
    int idx_ = -1;
    hipEvent_t a_{}, b_{};
  };

 private:
  struct Rec {
    int idx;
    hipEvent_t a, b;
  };
  hipEvent_t event() {
    if (!pool_.empty()) {
      hipEvent_t e = pool_.back();
      pool_.pop_back();
      return e;
    }
    hipEvent_t e;
    MSH_HIP(hipEventCreate(&e));
    return e;
  }
  void flush(hipStream_t s) {
    if (pending_.empty()) return;
    MSH_HIP(hipStreamSynchronize(s));
    for (auto& r : pending_) {
      float ms = 0.f;
      MSH_HIP(hipEventElapsedTime(&ms, r.a, r.b));
      entries_[r.idx].ms += ms;
      pool_.push_back(r.a);
      pool_.push_back(r.b);
    }
    pending_.clear();
  }
  bool on_ = false;
  std::vector<ProfEntry> entries_;
  std::map<std::string, int> index_;
  std::vector<Rec> pending_;
  std::vector<hipEvent_t> pool_;
};

}  // namespace msh
