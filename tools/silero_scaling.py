"""Host Silero VAD throughput vs thread count (each thread: its own 10 s clips through msh_host_silero_probabilities)."""
import ctypes as C, os, sys, tempfile, threading, time
import numpy as np
sys.path.insert(0, ".")
from moonshine_amd.hip_api import load_library
from moonshine_amd.synth import make_audio, make_silero_weights, save_safetensors
lib = load_library()
lib.msh_host_silero_probabilities.restype = C.c_int64
lib.msh_host_silero_probabilities.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_float)]
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "s.safetensors"); save_safetensors(p, make_silero_weights(2)); blob = open(p, "rb").read()
a = make_audio(5, 160000).astype(np.float32)
def worker(reps):
    probs = np.zeros(400, np.float32)
    for _ in range(reps):
        lib.msh_host_silero_probabilities(blob, len(blob), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], probs.ctypes.data_as(C.POINTER(C.c_float)), 400, None)
worker(1)
print("cpus visible", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
for nt in (1, 8, 16, 32, 64, 128, 256):
    reps = 6
    ts = [threading.Thread(target=worker, args=(reps,)) for _ in range(nt)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    clips = nt * reps
    print(f"threads {nt:4d}: {clips / dt:8.1f} clips/s = {clips * 10 / dt:9.0f} audio-s/s, {dt / reps * 1e3:7.1f} ms per clip per thread", flush=True)
