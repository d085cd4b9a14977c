"""One engine decodes (teacher-forced, graph mode) while another host thread keeps the GPU busy with ONE kind of kernel;
the decode's K/V cache is compared with its serial run.  Which co-running kernel disturbs the decode GEMMs?"""
import os, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moonshine_amd.hip_api import Engine, load_library
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

B, DEC = 256, 65
cfg = ARCHS["base"]
L, H, DH = cfg.dec_layers, cfg.heads, cfg.hidden // cfg.heads
with tempfile.TemporaryDirectory() as d:
    w = make_weights(cfg, 0)
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    e = Engine(0)
    e.load_weights_file(path)
lib = load_library()
audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
e.encode(device_ptrs=ptrs); e.synchronize()
toks, _ = e.decode(forced_steps=DEC)
teacher = np.asarray(toks, np.int32)
SMAX = (DEC + 7) // 8 * 8
def caches():
    return [e.debug_read(n).view(np.uint16).reshape(L, B, H, SMAX, DH)[:, :, :, :DEC].copy() for n in ("cache_k", "cache_v")]
e.decode(forced_steps=DEC, teacher=teacher)
ref = caches()
e.decode(forced_steps=DEC, teacher=teacher)
print("serial repeat identical:", all(np.array_equal(a, b) for a, b in zip(ref, caches())), flush=True)

xa = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
xb = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
big = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
side = torch.cuda.Stream()
def aggressor(kind, stop):
    with torch.cuda.stream(side):
        while not stop.is_set():
            if kind == "torch_matmul":
                for _ in range(20): torch.matmul(xa, xb)
                side.synchronize()
            elif kind == "torch_copy":
                for _ in range(20): big[: 128 << 20].copy_(big[128 << 20:])
                side.synchronize()
            elif kind == "dma_gemm":
                lib.msh_test_gemm_microbench(256, 32768, 416, 416, 0, 0, 400)
            elif kind == "dma_gemm_nodma":
                lib.msh_test_gemm_microbench(256, 32768, 416, 416, 0, 1, 400)
            elif kind == "dma_gemm_nostore":
                lib.msh_test_gemm_microbench(256, 32768, 416, 416, 0, 8, 400)
            elif kind == "astat_gemm":
                lib.msh_test_gemm_microbench(8192, 1664, 416, 416, 5, 0, 50)
            elif kind == "none":
                time.sleep(0.01)
for kind in os.environ.get("KINDS", "none,torch_matmul,torch_copy,dma_gemm,dma_gemm_nodma,dma_gemm_nostore,astat_gemm").split(","):
    stop = threading.Event()
    th = threading.Thread(target=aggressor, args=(kind, stop))
    th.start()
    time.sleep(0.05)
    out = []
    t0 = time.perf_counter()
    for rep in range(4):
        e.decode(forced_steps=DEC, teacher=teacher)
        got = caches()
        out.append(sum(int((g != r).sum()) for g, r in zip(got, ref)))
    dt = (time.perf_counter() - t0) / 4
    stop.set(); th.join()
    torch.cuda.synchronize()
    print(f"aggressor {kind:18s}: differing bf16 cache entries per decode {out}   ({dt*1e3:.0f} ms per decode+readback)", flush=True)
