#include "safetensors.h"

#include <string.h>

#include <cstdio>
#include <stdexcept>

namespace msh {
namespace {

// Tiny JSON reader for the safetensors header: objects, arrays, strings, integers.
struct JsonReader {
  const char* p;
  const char* end;
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
  }
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("safetensors header: ") + what); }
  void expect(char c) {
    ws();
    if (p >= end || *p != c) fail("unexpected character");
    ++p;
  }
  bool peek(char c) {
    ws();
    return p < end && *p == c;
  }
  std::string str() {
    ws();
    if (p >= end || *p != '"') fail("expected string");
    ++p;
    std::string out;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        ++p;
        if (p >= end) fail("bad escape");
        switch (*p) {
          case 'n': out.push_back('\n'); break;
          case 't': out.push_back('\t'); break;
          case 'r': out.push_back('\r'); break;
          case 'b': out.push_back('\b'); break;
          case 'f': out.push_back('\f'); break;
          case 'u': {
            if (end - p < 5) fail("bad \\u escape");
            unsigned v = 0;
            for (int i = 1; i <= 4; ++i) {
              char c = p[i];
              v <<= 4;
              if (c >= '0' && c <= '9') v |= c - '0';
              else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
              else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
              else fail("bad \\u escape");
            }
            p += 4;
            if (v < 0x80) out.push_back((char)v);
            else if (v < 0x800) {
              out.push_back((char)(0xC0 | (v >> 6)));
              out.push_back((char)(0x80 | (v & 0x3F)));
            } else {
              out.push_back((char)(0xE0 | (v >> 12)));
              out.push_back((char)(0x80 | ((v >> 6) & 0x3F)));
              out.push_back((char)(0x80 | (v & 0x3F)));
            }
            break;
          }
          default: out.push_back(*p);
        }
        ++p;
      } else {
        out.push_back(*p++);
      }
    }
    if (p >= end) fail("unterminated string");
    ++p;
    return out;
  }
  int64_t integer() {
    ws();
    bool neg = false;
    if (p < end && *p == '-') {
      neg = true;
      ++p;
    }
    if (p >= end || *p < '0' || *p > '9') fail("expected integer");
    int64_t v = 0;
    while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
    return neg ? -v : v;
  }
  std::vector<int64_t> int_array() {
    std::vector<int64_t> out;
    expect('[');
    if (peek(']')) {
      ++p;
      return out;
    }
    while (true) {
      out.push_back(integer());
      ws();
      if (peek(',')) {
        ++p;
        continue;
      }
      expect(']');
      return out;
    }
  }
  // skip any value (used for unknown keys)
  void skip() {
    ws();
    if (p >= end) fail("truncated");
    if (*p == '"') {
      str();
    } else if (*p == '{') {
      ++p;
      if (peek('}')) {
        ++p;
        return;
      }
      while (true) {
        str();
        expect(':');
        skip();
        if (peek(',')) {
          ++p;
          continue;
        }
        expect('}');
        return;
      }
    } else if (*p == '[') {
      ++p;
      if (peek(']')) {
        ++p;
        return;
      }
      while (true) {
        skip();
        if (peek(',')) {
          ++p;
          continue;
        }
        expect(']');
        return;
      }
    } else {
      while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
    }
  }
};

float half_to_float(uint16_t h) {
  uint32_t sign = (h >> 15) & 1, exp = (h >> 10) & 0x1f, man = h & 0x3ff, u;
  if (exp == 0) {
    if (man == 0) {
      u = sign << 31;
    } else {
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400));
      u = (sign << 31) | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
    }
  } else if (exp == 31) {
    u = (sign << 31) | 0x7f800000u | (man << 13);
  } else {
    u = (sign << 31) | ((exp + 112) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace

void SafeTensors::parse(const uint8_t* data, size_t size) {
  if (size < 8) throw std::runtime_error("safetensors: file too small");
  uint64_t hlen = 0;
  memcpy(&hlen, data, 8);
  if (hlen > size - 8) throw std::runtime_error("safetensors: header length exceeds file size");
  const uint8_t* blob = data + 8 + hlen;
  const size_t blob_size = size - 8 - hlen;
  JsonReader r{reinterpret_cast<const char*>(data + 8), reinterpret_cast<const char*>(data + 8 + hlen)};
  r.expect('{');
  if (r.peek('}')) return;
  while (true) {
    std::string key = r.str();
    r.expect(':');
    if (key == "__metadata__") {
      r.expect('{');
      if (r.peek('}')) {
        ++r.p;
      } else {
        while (true) {
          std::string k = r.str();
          r.expect(':');
          metadata[k] = r.str();
          if (r.peek(',')) {
            ++r.p;
            continue;
          }
          r.expect('}');
          break;
        }
      }
    } else {
      StTensor t;
      int64_t off0 = -1, off1 = -1;
      r.expect('{');
      while (true) {
        std::string k = r.str();
        r.expect(':');
        if (k == "dtype") t.dtype = r.str();
        else if (k == "shape") t.shape = r.int_array();
        else if (k == "data_offsets") {
          auto o = r.int_array();
          if (o.size() != 2) r.fail("data_offsets must have two entries");
          off0 = o[0];
          off1 = o[1];
        } else {
          r.skip();
        }
        if (r.peek(',')) {
          ++r.p;
          continue;
        }
        r.expect('}');
        break;
      }
      if (off0 < 0 || off1 < off0 || (uint64_t)off1 > blob_size)
        throw std::runtime_error("safetensors: bad data_offsets for " + key);
      t.data = blob + off0;
      t.nbytes = (size_t)(off1 - off0);
      size_t esz = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
      if (esz == 0) throw std::runtime_error("safetensors: unsupported dtype " + t.dtype + " for " + key);
      // model files are caller-supplied bytes: no negative extents, no product that wraps around
      uint64_t numel = 1;
      for (int64_t d : t.shape) {
        if (d < 0) throw std::runtime_error("safetensors: negative dimension in " + key);
        if (d != 0 && numel > (uint64_t)1 << 40) throw std::runtime_error("safetensors: shape of " + key + " is too large");
        numel *= (uint64_t)d;
      }
      if (numel > ((uint64_t)1 << 40) || numel * esz != (uint64_t)t.nbytes)
        throw std::runtime_error("safetensors: size mismatch for " + key);
      tensors[key] = t;
    }
    if (r.peek(',')) {
      ++r.p;
      continue;
    }
    r.expect('}');
    break;
  }
}

void SafeTensors::load_file(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  long n = -1;
  if (fseek(f, 0, SEEK_END) == 0) n = ftell(f);
  if (n < 0 || fseek(f, 0, SEEK_SET) != 0) {
    fclose(f);
    throw std::runtime_error("cannot determine the size of " + path);
  }
  owned.resize((size_t)n);
  size_t got = fread(owned.data(), 1, (size_t)n, f);
  fclose(f);
  if (got != (size_t)n) throw std::runtime_error("short read on " + path);
  parse(owned.data(), owned.size());
}

const StTensor& SafeTensors::get(const std::string& name) const {
  auto it = tensors.find(name);
  if (it == tensors.end()) throw std::runtime_error("safetensors: missing tensor " + name);
  used.insert(name);
  return it->second;
}

std::vector<std::string> SafeTensors::unused() const {
  std::vector<std::string> out;
  for (const auto& kv : tensors)
    if (used.count(kv.first) == 0) out.push_back(kv.first);
  return out;
}

std::vector<float> SafeTensors::to_f32(const std::string& name) const {
  const StTensor& t = get(name);
  std::vector<float> out((size_t)t.numel());
  if (t.dtype == "F32") {
    memcpy(out.data(), t.data, t.nbytes);
  } else if (t.dtype == "BF16") {
    const uint16_t* s = reinterpret_cast<const uint16_t*>(t.data);
    for (size_t i = 0; i < out.size(); ++i) {
      uint32_t u = ((uint32_t)s[i]) << 16;
      memcpy(&out[i], &u, 4);
    }
  } else if (t.dtype == "F16") {
    const uint16_t* s = reinterpret_cast<const uint16_t*>(t.data);
    for (size_t i = 0; i < out.size(); ++i) out[i] = half_to_float(s[i]);
  } else {
    throw std::runtime_error("safetensors: tensor " + name + " has dtype " + t.dtype + "; the engine reads F32, F16 and BF16 weights");
  }
  return out;
}

}  // namespace msh
