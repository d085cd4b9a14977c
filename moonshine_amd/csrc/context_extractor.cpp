#include "context_extractor.h"

#include <algorithm>
#include <map>

#include "host_utils.h"

namespace msh_host {

namespace {
// typographic punctuation real prose carries, folded to what the splitter understands
const char* const kFold[][2] = {
    {"\xe2\x80\x99", "'"}, {"\xe2\x80\x98", "'"}, {"\xe2\x80\x9c", " "}, {"\xe2\x80\x9d", " "},
    {"\xe2\x80\x93", " "}, {"\xe2\x80\x94", " "}, {"\xe2\x80\xa6", " "}, {"\xc2\xa0", " "},
};
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool is_letter(unsigned char c) { return c >= 0x80 || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
inline bool is_joiner(unsigned char c) { return c == '\'' || c == '-'; }

std::string strip_joiners(const std::string& w) {
  size_t b = 0, e = w.size();
  while (b < e && is_joiner((unsigned char)w[b])) ++b;
  while (e > b && is_joiner((unsigned char)w[e - 1])) --e;
  return w.substr(b, e - b);
}
size_t utf8_characters(const std::string& w) {
  size_t n = 0;
  for (const char c : w)
    if (((unsigned char)c & 0xC0) != 0x80) ++n;
  return n;
}
std::string lower_ascii(std::string w) {
  for (char& c : w)
    if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
  return w;
}
}  // namespace

std::string ContextExtractor::strip_possessive(const std::string& word) {
  const size_t n = word.size();
  if (n >= 2 && word[n - 2] == '\'' && (word[n - 1] == 's' || word[n - 1] == 'S')) return word.substr(0, n - 2);
  if (n >= 1 && word[n - 1] == '\'') return word.substr(0, n - 1);
  return word;
}

std::vector<std::string> ContextExtractor::candidate_words(const std::string& text) {
  std::string s = text;
  for (const auto& f : kFold) s = replace_all(s, f[0], f[1]);
  std::vector<std::string> words;
  std::string cur;
  auto flush = [&] {
    if (cur.empty()) return;
    const std::string w = strip_joiners(strip_possessive(strip_joiners(cur)));
    cur.clear();
    if (utf8_characters(w) < kMinCharacters) return;
    for (const char c : w)
      if (is_digit((unsigned char)c)) return;
    words.push_back(w);
  };
  for (const char ch : s) {
    const unsigned char c = (unsigned char)ch;
    if (is_letter(c) || is_digit(c) || (is_joiner(c) && !cur.empty()))
      cur.push_back(ch);
    else
      flush();
  }
  flush();
  return words;
}

std::vector<std::string> ContextExtractor::extract(const std::string& context, int32_t max_terms,
                                                   const SubwordCountFn& subword_count) {
  if (!subword_count) return {};
  const size_t limit = (size_t)(max_terms > 0 ? max_terms : kDefaultMaxTerms);
  const std::vector<std::string> words = candidate_words(context);
  struct Form {
    size_t count = 0, first = 0;
  };
  std::map<std::string, Form> forms;  // exact spellings
  for (size_t i = 0; i < words.size(); ++i) {
    auto it = forms.find(words[i]);
    if (it == forms.end()) it = forms.emplace(words[i], Form{0, i}).first;
    ++it->second.count;
  }
  struct Group {
    std::string term;
    size_t term_count = 0, first = 0, occurrences = 0, subwords = 0;
    bool set = false;
  };
  std::map<std::string, Group> groups;  // case variants share a group; the majority spelling (earliest on a tie) names it
  for (const auto& kv : forms) {
    Group& g = groups[lower_ascii(kv.first)];
    g.occurrences += kv.second.count;
    if (!g.set || kv.second.count > g.term_count || (kv.second.count == g.term_count && kv.second.first < g.first)) {
      g.term = kv.first;
      g.term_count = kv.second.count;
      g.first = kv.second.first;
      g.set = true;
    }
  }
  std::vector<Group> ranked;
  for (auto& kv : groups) {
    kv.second.subwords = subword_count(" " + kv.second.term);
    if (kv.second.subwords >= kMinSubwordTokens) ranked.push_back(kv.second);
  }
  std::sort(ranked.begin(), ranked.end(), [](const Group& a, const Group& b) {
    if (a.occurrences != b.occurrences) return a.occurrences > b.occurrences;
    if (a.subwords != b.subwords) return a.subwords > b.subwords;
    return a.first < b.first;
  });
  std::vector<std::string> terms;
  for (const Group& g : ranked) {
    if (terms.size() >= limit) break;
    terms.push_back(g.term);
  }
  return terms;
}

}  // namespace msh_host
