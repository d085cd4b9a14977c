// Minimal safetensors reader (header JSON + raw little-endian tensors).  The engine's native weight
// artifact is `model.safetensors` with HuggingFace Moonshine tensor names, so a
// real UsefulSensors/moonshine-* checkpoint loads unchanged.
#pragma once

#include <stdint.h>

#include <map>
#include <set>
#include <string>
#include <vector>

namespace msh {

struct StTensor {
  std::string dtype;  // "F32", "F16", "BF16"
  std::vector<int64_t> shape;
  const uint8_t* data = nullptr;
  size_t nbytes = 0;
  int64_t numel() const {
    int64_t n = 1;
    for (int64_t d : shape) n *= d;
    return n;
  }
};

struct SafeTensors {
  std::map<std::string, StTensor> tensors;
  std::map<std::string, std::string> metadata;
  std::vector<uint8_t> owned;  // backing store when loaded from a file
  mutable std::set<std::string> used;   // names handed out by get(): a loader can report what it never asked for

  // Parse an in-memory blob; tensor data pointers alias `data` (must outlive this object).
  void parse(const uint8_t* data, size_t size);
  void load_file(const std::string& path);
  const StTensor& get(const std::string& name) const;
  bool has(const std::string& name) const { return tensors.count(name) != 0; }
  // Convert tensor to fp32 (from F32 / F16 / BF16; any other dtype is an error, not a reinterpretation).
  std::vector<float> to_f32(const std::string& name) const;
  // tensors of the file nobody asked for
  std::vector<std::string> unused() const;
};

}  // namespace msh
