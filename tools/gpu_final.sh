#!/bin/bash
# Round-end evidence run: GPU tests, smoke, default bench, rocprofv3 kernel stats of the same command, and HBM-byte
# counters (separate --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) for the decode kernels.
set -u
export MSH_DEV_KNOBS=1   # the library reads its developer switches only with this set
TAG=${1:-final}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json
# the launch line the driver uses for N > 1, here with one rank: rendezvous, RCCL init, scatter / gather / max-over-ranks on hardware
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 6 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api > gpurun_out/${TAG}_bench_torchrun1.json 2> gpurun_out/${TAG}_bench_torchrun1.err; echo "torchrun rc=$?"; cut -c1-200 gpurun_out/${TAG}_bench_torchrun1.json
# kernel stats twice: steps strictly serial (the per-kernel durations the roofline figures are about), and the default
# command with batches in flight (the same kernels stretched by the overlap)
CMD="python $R/bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- $CMD > /tmp/prof_${TAG}.log 2>&1)
f=$(find /tmp/prof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats_b256_serial.csv && head -8 "$f" | cut -c1-160
CMD2="python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_${TAG} -- $CMD2 > /tmp/prof2_${TAG}.log 2>&1)
f=$(find /tmp/prof2_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats_b256_inflight.csv && head -4 "$f" | cut -c1-160
f=$(find /tmp/prof2_${TAG} -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" gpurun_out/${TAG}_kernel_trace_inflight.csv.gz <<'PYT'
import csv, gzip, re, sys
with open(sys.argv[1]) as src, gzip.open(sys.argv[2], "wt") as dst:
    w = csv.writer(dst)
    w.writerow(["queue", "start_ns", "end_ns", "kernel"])
    for r in csv.DictReader(src):
        name = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:80]
        w.writerow([r["Queue_Id"], r["Start_Timestamp"], r["End_Timestamp"], name])
PYT
pass() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python $R/bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8 > /tmp/pmc_${TAG}_$name.log 2>&1)
  tail -1 /tmp/pmc_${TAG}_$name.log | cut -c1-160
}
pass fetch FETCH_SIZE
# (rocprofv3 itself has crashed in this pass on some boxes: one retry)
[ -z "$(find /tmp/pmc_${TAG}_fetch -name '*counter_collection.csv' 2>/dev/null | head -1)" ] && pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
[ -x tools/build/dec_gemm_timeline ] && tools/build/dec_gemm_timeline > gpurun_out/${TAG}_dec_gemm_timeline.txt 2>&1
[ -x tools/build/launch_floor ] && tools/build/launch_floor > gpurun_out/${TAG}_launch_floor.txt 2>&1
python - <<PY
import csv, glob, collections, json, re
# MFMA busy normalised to the chip: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs) / (kernel duration in cycles x 1024 SIMDs);
# the duration is GRBM_GUI_ACTIVE, which this stack reports summed over the 8 XCDs (a 0.27 ms kernel reads 5.3 M): / 8
files = glob.glob("/tmp/pmc_${TAG}_mfma/**/*counter_collection.csv", recursive=True)
if files:
    agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(files[0])):
        k = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:90]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    rows = {}
    for k, n in cnt.items():
        gui = agg.get((k, "GRBM_GUI_ACTIVE"), 0.0) / 8.0
        mf = agg.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0)
        wave = max(agg.get((k, "SQ_WAVE_CYCLES"), 0.0), 1.0)
        rows[k] = {"dispatches": n, "duration_cycles_per_dispatch": round(gui / n, 1),
                   "mfma_busy_frac_of_chip": round(mf / (gui * 1024.0), 4) if gui > 0 else None,
                   "valu_active_over_wave": round(agg.get((k, "SQ_ACTIVE_INST_VALU"), 0.0) / wave, 3),
                   "wait_any_over_wave": round(agg.get((k, "SQ_WAIT_ANY"), 0.0) / wave, 3)}
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE, "
                     "bench.py --in-flight 1 --steps 1 --warmup 0 (B=256, base).  mfma_busy_frac_of_chip = MFMA busy cycles / (kernel "
                     "cycles x 1024 SIMDs): the fraction of the chip's matrix-pipe time the kernel uses (1.0 = every SIMD issuing MFMAs "
                     "back to back).  GRBM_GUI_ACTIVE is summed over the 8 XCDs on this stack, hence / 8.",
           "kernels": dict(sorted(rows.items(), key=lambda kv: -(kv[1]["duration_cycles_per_dispatch"] * kv[1]["dispatches"])))}
    json.dump(out, open("gpurun_out/${TAG}_pmc_mfma_busy.json", "w"), indent=1)
    for k, v in list(out["kernels"].items())[:14]:
        print("%-86s n=%5d mfma_busy=%s valu=%.2f wait=%.2f" % (k[:86], v["dispatches"], v["mfma_busy_frac_of_chip"], v["valu_active_over_wave"], v["wait_any_over_wave"]))
PY
python - <<PY
import csv, glob, collections, json, re
GROUPS = {  # bench.py kernel group -> kernel-name pattern
    "dec_cross_attention": r"dec_cross_absorbed_kernel|dec_cross_attention_kernel", "dec_self_attention": r"dec_self_attention_kernel",
    "dec_crossq_gemm": r"dec_crossq2_kernel|EpiQtFrag", "dec_ctx_resid_gemm": r"gemm_dec_kernel<(104|72), ",
    "dec_fc2_resid_gemm": r"gemm_dec_kernel<(52|36), \d+, false, .*EpiDecResidFm<true>", "dec_proj_resid_gemm": r"gemm_dec_kernel<(13|9), \d+, false, .*EpiDecResidFm<false>",
    "enc_attention": r"enc_attention_res_kernel", "dec_qkv_gemm": r"gemm_dec_kernel<\d+, \d+, true, .*EpiDecQkv",
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

# streaming workload (BASELINE config 5): kernel stats of one timed step
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sprof_${TAG} -o s -- python $R/bench.py --workload streaming --steps 1 --warmup 1 --no-stream-profile > $R/gpurun_out/${TAG}_stream_traced.json 2> /dev/null)
f=$(find /tmp/sprof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats_streaming.csv && head -6 "$f" | cut -c1-160
timeout 300 python tools/mlp_microbench.py > gpurun_out/${TAG}_mlp_fused_ablations.txt 2>&1; tail -3 gpurun_out/${TAG}_mlp_fused_ablations.txt
timeout 300 python tools/panel_microbench.py > gpurun_out/${TAG}_qkv_panel_ablations.txt 2>&1; head -3 gpurun_out/${TAG}_qkv_panel_ablations.txt
# round 5: the single-clip latency path (p50 + chain costs + the clip's encoder kernel by kernel), chain costs at small batches,
# the fused-kernel probes with the shader clock, the parity margins recorded by the suite above
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_latency_batch1.txt; head -1 gpurun_out/${TAG}_latency_batch1.txt | cut -c1-200
timeout 300 python tools/chain_probe.py 1 2 4 8 16 32 63 2>&1 | grep "^B=" > gpurun_out/${TAG}_small_batch_chain_costs.txt
[ -x tools/build/mfma_peak ] && timeout 120 tools/build/mfma_peak > gpurun_out/${TAG}_mfma_issue_ceiling_with_clock.txt 2>&1
cp gpurun_out/parity_margins.json gpurun_out/${TAG}_parity_margins.json 2>/dev/null
# round 6: the MFMA + LDS-DMA loop at one and two waves per SIMD, and the streaming AR step kernel group by kernel group
[ -x tools/build/mfma_dma_overlap ] && timeout 120 tools/build/mfma_dma_overlap > gpurun_out/${TAG}_mfma_dma_overlap.txt 2>&1
MASKS="0x3ff 0x1 0x2 0x4 0x8 0x10 0x20 0x40 0x80 0x100 0x0" bash tools/gpu_stream_chain.sh ${TAG} > /dev/null 2>&1; cat gpurun_out/${TAG}_stream_chain_costs.txt | cut -c1-120
