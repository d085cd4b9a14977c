#!/bin/bash
# HBM-byte counters of the decode / encoder kernels alone (separate rocprofv3 --pmc passes, no fp8 sub-run): the part of
# tools/gpu_final.sh that writes <tag>_pmc_traffic.json
set -u
TAG=${1:-pmc}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
pass() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python $R/bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > /tmp/pmc_${TAG}_$name.log 2>&1)
  tail -1 /tmp/pmc_${TAG}_$name.log | cut -c1-160
}
pass fetch FETCH_SIZE
# (rocprofv3 itself has crashed in this pass on some boxes: one retry)
[ -z "$(find /tmp/pmc_${TAG}_fetch -name '*counter_collection.csv' 2>/dev/null | head -1)" ] && pass fetch FETCH_SIZE
pass write WRITE_SIZE
python - <<PY
import csv, glob, collections, json, re
GROUPS = {  # bench.py kernel group -> kernel-name pattern
    "dec_cross_attention": r"dec_cross_absorbed_kernel|dec_cross_attention_kernel", "dec_self_attention": r"dec_self_attention_kernel",
    "dec_crossq_gemm": r"dec_crossq2_kernel|EpiQtFrag", "dec_ctx_resid_gemm": r"gemm_dec_kernel<(104|72), ",
    "dec_fc2_resid_gemm": r"gemm_dec_kernel<(52|36), \d+, false, .*EpiDecResidFm<true>", "dec_proj_resid_gemm": r"gemm_dec_kernel<(13|9), \d+, false, .*EpiDecResidFm<false>",
    "enc_attention": r"enc_attention_kernel", "dec_qkv_gemm": r"gemm_dec_kernel<\d+, \d+, true, .*EpiDecQkv",
    "dec_fc1_swiglu_gemm": r"gemm_dec_kernel<\d+, \d+, true, .*EpiSwiGLU", "conv2_gelu_gemm": r"EpiGnBiasGeluBf16", "enc_fc1_gelu_gemm": r"gemm_astat_kernel.*EpiBiasGeluBf16",
    "enc_oproj_mlp_fused": r"mlp_fused_kernel", "enc_qkv_panel": r"panel_gemm_kernel", "cross_kv_gemm": r"EpiCrossKV",
}
out = {}
for name in ("fetch", "write"):
    files = glob.glob(f"/tmp/pmc_${TAG}_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter file for", name); continue
    agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for (k, c), v in agg.items():
        out.setdefault(k, {"dispatches": cnt[k]})[c] = v
res = {"cross_attention": "absorbed" if any("dec_cross_absorbed_kernel" in k for k in out) else "kv",
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0, B=256; "
                 "units KiB summed over dispatches; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md) "
                 "in traffic_bytes_per_launch; WRITE_SIZE uncalibrated", "groups": {}}
for g, pat in GROUPS.items():
    ks = [k for k in out if re.search(pat, k)]
    if not ks: continue
    n = sum(out[k]["dispatches"] for k in ks)
    fetch = sum(out[k].get("FETCH_SIZE", 0.0) for k in ks); write = sum(out[k].get("WRITE_SIZE", 0.0) for k in ks)
    res["groups"][g] = {"dispatches": n, "fetch_kib_per_launch": fetch / n, "write_kib_per_launch": write / n,
                        "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / n}
    print(g, res["groups"][g])
json.dump(res, open("gpurun_out/${TAG}_pmc_traffic.json", "w"), indent=1)
PY

