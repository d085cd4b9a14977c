#include "context_biaser.h"

#include <math.h>

#include <algorithm>

#include "host_utils.h"

namespace msh_host {

void ContextBiaser::add_token_sequence(const std::vector<int32_t>& tokens) {
  if (tokens.empty()) return;
  int32_t node = 0;
  for (const int32_t t : tokens) {
    auto it = nodes_[node].children.find(t);
    if (it != nodes_[node].children.end()) {
      node = it->second;
      continue;
    }
    const int child_depth = nodes_[node].depth + 1;
    const int32_t child = (int32_t)nodes_.size();
    nodes_[node].children.emplace(t, child);
    nodes_.push_back(Node{});
    nodes_[child].depth = child_depth;
    max_depth_ = std::max(max_depth_, child_depth);
    node = child;
  }
  ++sequence_count_;
}

std::vector<std::string> ContextBiaser::variants_for_term(const std::string& term) {
  const std::string t = trim(term);  // spaces and tabs, like the reference string-utils trim
  if (t.empty()) return {};
  if (t.compare(0, 3, "\xe2\x96\x81") == 0) return {t};  // already anchored to a word start by the caller
  return {t, " " + t};
}

void ContextBiaser::clear() {
  nodes_.assign(1, Node{});
  sequence_count_ = 0;
  max_depth_ = 0;
  reset();
}

float ContextBiaser::bonus_for_depth(int depth) const {
  if (depth <= 0) return 0.0f;
  return boost_ * (1.0f + logf((float)depth));
}

void ContextBiaser::apply(float* logits, int vocab_size) const {
  if (logits == nullptr || sequence_count_ == 0) return;
  std::vector<std::pair<int32_t, float>> pending;
  for (const int32_t n : active_) {
    const float bonus = bonus_for_depth(nodes_[n].depth + 1);
    for (const auto& c : nodes_[n].children) {
      if (c.first < 0 || c.first >= vocab_size) continue;
      bool found = false;
      for (auto& p : pending)
        if (p.first == c.first) {
          p.second = std::max(p.second, bonus);
          found = true;
          break;
        }
      if (!found) pending.emplace_back(c.first, bonus);
    }
  }
  for (const auto& p : pending) logits[p.first] += p.second;
}

void ContextBiaser::advance(int32_t token) {
  if (sequence_count_ == 0) return;
  std::vector<int32_t> next{0};
  for (const int32_t n : active_) {
    auto it = nodes_[n].children.find(token);
    if (it != nodes_[n].children.end()) next.push_back(it->second);
  }
  active_.swap(next);
}

ContextBiaser::Flat ContextBiaser::flatten() const {
  Flat f;
  f.child_off.push_back(0);
  for (const Node& n : nodes_) {
    for (const auto& c : n.children) {  // std::map: ascending token order
      f.child_tok.push_back(c.first);
      f.child_node.push_back(c.second);
    }
    f.child_off.push_back((int32_t)f.child_tok.size());
    f.depth.push_back(n.depth);
  }
  for (int d = 0; d <= max_depth_ + 1; ++d) f.depth_bonus.push_back(bonus_for_depth(d));
  return f;
}

}  // namespace msh_host
