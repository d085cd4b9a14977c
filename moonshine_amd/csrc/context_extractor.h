// Key terms out of a free-form passage (reference core/context-extractor.{h,cpp}): words the loaded tokenizer needs
// two or more subwords for, ranked by how often the passage says them.  Feeds Transcriber::set_context.
#pragma once

#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

namespace msh_host {

class ContextExtractor {
 public:
  static constexpr int32_t kDefaultMaxTerms = 200;  // reference core/context-extractor.h:36
  static constexpr size_t kMinSubwordTokens = 2;    // :43
  static constexpr size_t kMinCharacters = 3;       // :48
  using SubwordCountFn = std::function<size_t(const std::string& word)>;

  static std::vector<std::string> extract(const std::string& context, int32_t max_terms, const SubwordCountFn& subword_count);
  static std::vector<std::string> candidate_words(const std::string& text);
  static std::string strip_possessive(const std::string& word);
};

}  // namespace msh_host
