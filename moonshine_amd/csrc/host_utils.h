// Small host-side helpers shared by the transcriber layer: option-value parsing, logging,
// sample-rate conversion and the 16-bit PCM WAV reader.  Semantics follow the reference helpers they
// replace (cited per function); the code is written for this library.
#pragma once

#include <stdint.h>

#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace msh_host {

constexpr int32_t kSampleRate = 16000;  // every model consumes 16 kHz mono

// stderr log line with file:line, the shape of the reference's LOGF (core/moonshine-utils/debug-utils.h:33-43)
#define MSH_LOGF(fmt, ...) std::fprintf(stderr, "[moonshine %s:%d] " fmt "\n", __FILE__, __LINE__, ##__VA_ARGS__)

std::string to_lower(const std::string& s);
std::string trim(const std::string& s, const std::string& whitespace = " \t");  // string-utils.cpp:21-29
std::vector<std::string> split(const std::string& s, const std::string& delim);
std::string replace_all(std::string s, const std::string& from, const std::string& to);  // string-utils.cpp:9-17

// Option values are strings (reference core/moonshine-utils/string-utils.cpp:92-161): bools accept only
// true/false/1/0 (any case); numbers go through stof / stoi / stoul; anything else throws.
bool parse_bool(const std::string& v);
float parse_float(const std::string& v);
int32_t parse_int32(const std::string& v);
size_t parse_size(const std::string& v);

// Box-filter decimation / linear interpolation (reference core/resampler.cpp:5-86); identity when the
// rates match.
std::vector<float> resample(const std::vector<float>& audio, float in_rate, float out_rate);

// 16-bit PCM WAV -> float / 32768, channel count ignored (reference core/moonshine-utils/debug-utils.cpp:52-190)
bool load_wav(const std::string& path, std::vector<float>* samples, int32_t* sample_rate);
bool save_wav(const std::string& path, const float* samples, size_t count, int32_t sample_rate);

std::string join_path(const std::string& dir, const std::string& name);
bool file_exists(const std::string& path);
bool read_file(const std::string& path, std::vector<uint8_t>* out);

// fn(i) for i in [0, n) on up to max_threads host threads (0 = one per hardware thread); items are handed out one at a
// time, the first exception is re-thrown on the caller's thread after every worker has stopped
void parallel_for(size_t n, const std::function<void(size_t)>& fn, unsigned max_threads = 0);
// CPUs this process may actually use: the affinity mask, cut down to the cgroup CPU quota when there is one (cgroup v2
// cpu.max or v1 cpu.cfs_quota_us / cpu.cfs_period_us).  A container with 256 visible cores and a 16-CPU quota gets 16:
// more threads than that only buy throttling (measured on the benchmark box: the host VAD peaks at 2x the quota and
// loses half its rate at 8x).
unsigned effective_cpus();
// Several GPUs in one process (load options num_gpus / devices): the calling thread is restricted to the CPUs of the NUMA node
// the GPU with PCI bus id `pci_bus_id` ("0000:c1:00.0") hangs off, intersected with the thread's current mask -- the lanes'
// host threads and the pinned staging blocks they first-touch then sit next to their GPU instead of wherever the scheduler
// put them (an 8-GPU host moves 20-40 GB/s of PCM through those blocks).  No-op (returns false) when sysfs does not name a
// node, the intersection is empty or MSH_PIN_CPUS=0.
bool pin_thread_to_gpu_node(const char* pci_bus_id);
// the CPU list of "/sys/devices/system/node/node<N>/cpulist" syntax ("0-63,128-191") as a vector (exposed for the CPU test)
std::vector<int> parse_cpu_list(const std::string& text);

}  // namespace msh_host
