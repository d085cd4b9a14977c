// Contextual biasing towards caller-supplied key terms: the reference's ContextBiaser
// (core/context-biaser.{h,cpp}) as a host-side trie builder.  The reference applies the bonuses on the host inside
// its decode loop; here token choice happens on the device, so this class only compiles the key terms into a flat,
// sorted trie (`Flat`) that the streaming engine walks in a kernel (k_stream.hip: bias_rows_kernel).  apply / advance
// exist for tests and keep the reference's semantics: the root is always active, a token proposed by several
// active nodes gets the largest bonus once, bonus(depth) = boost * (1 + ln depth).
#pragma once

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

namespace msh_host {

class ContextBiaser {
 public:
  static constexpr float kDefaultBoost = 2.0f;  // reference core/context-biaser.h:43

  void add_token_sequence(const std::vector<int32_t>& tokens);
  static std::vector<std::string> variants_for_term(const std::string& term);
  void set_boost(float boost) { boost_ = boost; }
  float boost() const { return boost_; }
  bool empty() const { return sequence_count_ == 0; }
  size_t sequence_count() const { return sequence_count_; }
  int max_depth() const { return max_depth_; }
  void clear();
  void reset() { active_.assign(1, 0); }
  void apply(float* logits, int vocab_size) const;
  void advance(int32_t token);
  float bonus_for_depth(int depth) const;

  // children of node n = [child_off[n], child_off[n+1]) in child_tok / child_node, sorted by token
  struct Flat {
    std::vector<int32_t> child_off, child_tok, child_node, depth;
    std::vector<float> depth_bonus;  // indexed by depth, max_depth + 2 entries
  };
  Flat flatten() const;

 private:
  struct Node {
    std::map<int32_t, int32_t> children;
    int depth = 0;
  };
  std::vector<Node> nodes_{Node{}};
  std::vector<int32_t> active_{0};
  int max_depth_ = 0;
  size_t sequence_count_ = 0;
  float boost_ = kDefaultBoost;
};

}  // namespace msh_host
