#include "host_utils.h"

#include <ctype.h>
#include <pthread.h>
#include <sched.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace msh_host {

std::string to_lower(const std::string& s) {
  std::string out = s;
  for (char& c : out)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  return out;
}

std::string trim(const std::string& s, const std::string& whitespace) {
  const size_t a = s.find_first_not_of(whitespace);
  if (a == std::string::npos) return "";
  const size_t b = s.find_last_not_of(whitespace);
  return s.substr(a, b - a + 1);
}

std::vector<std::string> split(const std::string& s, const std::string& delim) {
  std::vector<std::string> out;
  if (delim.empty()) {
    out.push_back(s);
    return out;
  }
  size_t pos = 0;
  while (true) {
    const size_t hit = s.find(delim, pos);
    if (hit == std::string::npos) {
      out.push_back(s.substr(pos));
      return out;
    }
    out.push_back(s.substr(pos, hit - pos));
    pos = hit + delim.size();
  }
}

std::string replace_all(std::string s, const std::string& from, const std::string& to) {
  if (from.empty()) return s;
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) {
    s.replace(pos, from.size(), to);
    pos += to.size();
  }
  return s;
}

bool parse_bool(const std::string& v) {
  const std::string l = to_lower(v);
  if (l == "true" || l == "1") return true;
  if (l == "false" || l == "0") return false;
  throw std::runtime_error("Invalid boolean string: '" + v + "'");
}

float parse_float(const std::string& v) {
  try {
    return std::stof(v);
  } catch (const std::exception& e) {
    throw std::runtime_error("Invalid float string: '" + v + "': " + e.what());
  }
}

int32_t parse_int32(const std::string& v) {
  try {
    return (int32_t)std::stoi(v, nullptr, 10);
  } catch (const std::exception& e) {
    throw std::runtime_error("Invalid int32_t string: '" + v + "': " + e.what());
  }
}

size_t parse_size(const std::string& v) {
  try {
    return (size_t)std::stoul(v, nullptr, 10);
  } catch (const std::exception& e) {
    throw std::runtime_error("Invalid size_t string: '" + v + "': " + e.what());
  }
}

// reference core/resampler.cpp: output length = in_len * out_rate / in_rate (float arithmetic, truncated);
// decimation averages the input samples [floor(i*r), floor((i+1)*r)] inclusive; interpolation is linear
// with the last sample held.
std::vector<float> resample(const std::vector<float>& audio, float in_rate, float out_rate) {
  if (in_rate == out_rate) return audio;
  const size_t n_in = audio.size();
  const size_t n_out = (size_t)(n_in * out_rate / in_rate);
  std::vector<float> out(n_out);
  const float ratio = in_rate / out_rate;
  if (in_rate > out_rate) {
    for (size_t i = 0; i < n_out; ++i) {
      const size_t lo = (size_t)(i * ratio);
      size_t hi = (size_t)((i + 1) * ratio);
      if (hi >= n_in) hi = n_in - 1;
      float sum = 0.f;
      size_t cnt = 0;
      for (size_t j = lo; j <= hi; ++j) {
        sum += audio[j];
        ++cnt;
      }
      out[i] = cnt ? sum / cnt : 0.f;
    }
  } else {
    for (size_t i = 0; i < n_out; ++i) {
      const float pos = i * ratio;
      const size_t idx = (size_t)pos;
      const float frac = pos - idx;
      if (idx >= n_in - 1) {
        out[i] = audio[n_in - 1];
      } else {
        out[i] = audio[idx] + frac * (audio[idx + 1] - audio[idx]);
      }
    }
  }
  return out;
}

namespace {
struct File {
  FILE* f;
  explicit File(const char* path, const char* mode) : f(std::fopen(path, mode)) {}
  ~File() {
    if (f) std::fclose(f);
  }
};
template <class T>
bool rd(FILE* f, T* v) {
  return std::fread(v, sizeof(T), 1, f) == 1;
}
// scan RIFF chunks until `id`; leaves the file positioned at the chunk payload
bool seek_chunk(FILE* f, const char* id, uint32_t* size) {
  char cid[4];
  while (std::fread(cid, 1, 4, f) == 4) {
    if (!rd(f, size)) return false;
    if (memcmp(cid, id, 4) == 0) return true;
    std::fseek(f, *size, SEEK_CUR);
  }
  return false;
}
}  // namespace

bool load_wav(const std::string& path, std::vector<float>* samples, int32_t* sample_rate) {
  samples->clear();
  File file(path.c_str(), "rb");
  if (!file.f) return false;
  char tag[4];
  if (std::fread(tag, 1, 4, file.f) != 4 || memcmp(tag, "RIFF", 4) != 0) return false;
  std::fseek(file.f, 4, SEEK_CUR);
  if (std::fread(tag, 1, 4, file.f) != 4 || memcmp(tag, "WAVE", 4) != 0) return false;
  uint32_t size = 0;
  if (!seek_chunk(file.f, "fmt ", &size) || size < 16) return false;
  uint16_t format = 0, channels = 0, align = 0, bits = 0;
  uint32_t rate = 0, byte_rate = 0;
  if (!rd(file.f, &format) || !rd(file.f, &channels) || !rd(file.f, &rate) || !rd(file.f, &byte_rate) ||
      !rd(file.f, &align) || !rd(file.f, &bits))
    return false;
  if (size > 16) std::fseek(file.f, size - 16, SEEK_CUR);
  if (format != 1 || bits != 16) return false;  // only 16-bit PCM
  if (!seek_chunk(file.f, "data", &size)) return false;
  const long start = std::ftell(file.f);
  std::fseek(file.f, 0, SEEK_END);
  const long end = std::ftell(file.f);
  std::fseek(file.f, start, SEEK_SET);
  if (end < start) return false;
  size_t bytes = std::min<size_t>(size, (size_t)(end - start));
  const size_t n = bytes / 2;
  if (n == 0) return false;
  std::vector<int16_t> raw(n);
  if (std::fread(raw.data(), 2, n, file.f) != n) return false;
  samples->resize(n);
  for (size_t i = 0; i < n; ++i) (*samples)[i] = (float)raw[i] / 32768.0f;  // interleaved channels stay interleaved
  *sample_rate = (int32_t)rate;
  return true;
}

bool save_wav(const std::string& path, const float* samples, size_t count, int32_t sample_rate) {
  File file(path.c_str(), "wb");
  if (!file.f) return false;
  const uint32_t data_bytes = (uint32_t)(count * 2), riff = 36 + data_bytes, fmt_size = 16, rate = (uint32_t)sample_rate,
                 byte_rate = rate * 2;
  const uint16_t format = 1, channels = 1, align = 2, bits = 16;
  std::fwrite("RIFF", 1, 4, file.f);
  std::fwrite(&riff, 4, 1, file.f);
  std::fwrite("WAVEfmt ", 1, 8, file.f);
  std::fwrite(&fmt_size, 4, 1, file.f);
  std::fwrite(&format, 2, 1, file.f);
  std::fwrite(&channels, 2, 1, file.f);
  std::fwrite(&rate, 4, 1, file.f);
  std::fwrite(&byte_rate, 4, 1, file.f);
  std::fwrite(&align, 2, 1, file.f);
  std::fwrite(&bits, 2, 1, file.f);
  std::fwrite("data", 1, 4, file.f);
  std::fwrite(&data_bytes, 4, 1, file.f);
  for (size_t i = 0; i < count; ++i) {
    float v = samples[i] * 32768.0f;
    v = v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v);
    const int16_t s = (int16_t)v;
    std::fwrite(&s, 2, 1, file.f);
  }
  return true;
}

std::string join_path(const std::string& dir, const std::string& name) {
  if (dir.empty()) return name;
  if (dir.back() == '/') return dir + name;
  return dir + "/" + name;
}

bool file_exists(const std::string& path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0;
}

bool read_file(const std::string& path, std::vector<uint8_t>* out) {
  File file(path.c_str(), "rb");
  if (!file.f) return false;
  std::fseek(file.f, 0, SEEK_END);
  const long n = std::ftell(file.f);
  std::fseek(file.f, 0, SEEK_SET);
  if (n < 0) return false;
  out->resize((size_t)n);
  return std::fread(out->data(), 1, (size_t)n, file.f) == (size_t)n;
}

std::vector<int> parse_cpu_list(const std::string& text) {
  std::vector<int> out;
  size_t i = 0;
  while (i < text.size()) {
    while (i < text.size() && !isdigit((unsigned char)text[i])) ++i;
    if (i >= text.size()) break;
    int a = 0;
    while (i < text.size() && isdigit((unsigned char)text[i])) a = a * 10 + (text[i++] - '0');
    int b = a;
    if (i < text.size() && text[i] == '-') {
      ++i;
      b = 0;
      while (i < text.size() && isdigit((unsigned char)text[i])) b = b * 10 + (text[i++] - '0');
    }
    for (int c = a; c <= b && c < 4096; ++c) out.push_back(c);
  }
  return out;
}

bool pin_thread_to_gpu_node(const char* pci_bus_id) {
  const char* off = getenv("MSH_PIN_CPUS");
  if (off != nullptr && off[0] == '0') return false;
  if (pci_bus_id == nullptr || pci_bus_id[0] == 0) return false;
  std::string id(pci_bus_id);
  for (char& c : id) c = (char)tolower((unsigned char)c);
  auto slurp = [](const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "r");
    if (f == nullptr) return false;
    char buf[4096];
    const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[got] = 0;
    *out = buf;
    return got > 0;
  };
  std::string text;
  if (!slurp("/sys/bus/pci/devices/" + id + "/numa_node", &text)) return false;
  const int node = atoi(text.c_str());
  if (node < 0) return false;
  if (!slurp("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", &text)) return false;
  cpu_set_t cur, want;
  CPU_ZERO(&cur);
  CPU_ZERO(&want);
  if (pthread_getaffinity_np(pthread_self(), sizeof(cur), &cur) != 0) return false;
  int n = 0;
  for (int c : parse_cpu_list(text))
    if (c < CPU_SETSIZE && CPU_ISSET(c, &cur)) {
      CPU_SET(c, &want);
      ++n;
    }
  if (n == 0) return false;
  return pthread_setaffinity_np(pthread_self(), sizeof(want), &want) == 0;
}

unsigned effective_cpus() {
  static const unsigned cached = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    auto read_file = [](const char* path, char* buf, size_t cap) -> bool {
      FILE* f = fopen(path, "r");
      if (f == nullptr) return false;
      const size_t got = fread(buf, 1, cap - 1, f);
      fclose(f);
      buf[got] = 0;
      return got > 0;
    };
    char buf[128];
    double quota = -1.0, period = -1.0;
    if (read_file("/sys/fs/cgroup/cpu.max", buf, sizeof(buf))) {           // cgroup v2: "<quota|max> <period>"
      if (strncmp(buf, "max", 3) != 0) sscanf(buf, "%lf %lf", &quota, &period);
    } else if (read_file("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", buf, sizeof(buf))) {   // cgroup v1
      quota = atof(buf);
      if (read_file("/sys/fs/cgroup/cpu/cpu.cfs_period_us", buf, sizeof(buf))) period = atof(buf);
    }
    if (quota > 0.0 && period > 0.0) {
      const unsigned q = (unsigned)ceil(quota / period);
      if (q >= 1 && q < n) n = q;
    }
    return n;
  }();
  return cached;
}

void parallel_for(size_t n, const std::function<void(size_t)>& fn, unsigned max_threads) {
  unsigned nt = max_threads ? max_threads : effective_cpus();
  if (nt == 0) nt = 1;
  if (nt > n) nt = (unsigned)n;
  if (nt <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  std::exception_ptr err;
  std::mutex err_mutex;
  auto work = [&] {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n || failed.load()) return;
      try {
        fn(i);
      } catch (...) {
        std::lock_guard<std::mutex> lock(err_mutex);
        if (!err) err = std::current_exception();
        failed.store(true);
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(nt - 1);
  for (unsigned t = 0; t + 1 < nt; ++t) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  if (err) std::rethrow_exception(err);
}

}  // namespace msh_host
