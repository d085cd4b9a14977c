#!/bin/bash
# rocprofv3 counter passes over the default bench workload (one step): SQ activity / MFMA busy / LDS conflicts, then
# FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: TCC slots do not fit together).
set -u
TAG=${1:-pmc2}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency"
pass() { local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- $CMD > /tmp/pmc_${TAG}_$name.log 2>&1)
  tail -1 /tmp/pmc_${TAG}_$name.log | cut -c1-120; }
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python - <<PY
import csv, glob, collections, json, re
out = {}
for name in ("sq", "fetch", "write"):
    files = glob.glob(f"/tmp/pmc_${TAG}_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter file for", name); continue
    agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(files[0])):
        k = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:96]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for (k, c), v in agg.items():
        out.setdefault(k, {"dispatches": cnt[k]})[c] = v
summary = {"source": "rocprofv3 --pmc, bench.py --steps 1 --warmup 0 (B=256, base); per kernel, summed over dispatches; "
                     "FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE counts half the bytes of wide coalesced reads on gfx950)", "kernels": {}}
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    d = v["dispatches"]; wave = max(v.get("SQ_WAVE_CYCLES", 0), 1); busy = max(v.get("SQ_BUSY_CYCLES", 0), 1)
    summary["kernels"][k] = {
        "dispatches": d,
        "mfma_busy_over_sq_busy": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy, 4),
        "wait_any_over_wave": round(v.get("SQ_WAIT_ANY", 0) / wave, 3),
        "wait_inst_over_wave": round(v.get("SQ_WAIT_INST_ANY", 0) / wave, 3),
        "active_over_wave": round(v.get("SQ_ACTIVE_INST_ANY", 0) / wave, 3),
        "lds_conflict_over_lds_active": round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1), 3),
        "fetch_kib_per_dispatch": round(v.get("FETCH_SIZE", 0) / d, 1), "write_kib_per_dispatch": round(v.get("WRITE_SIZE", 0) / d, 1)}
json.dump(summary, open("gpurun_out/${TAG}_pmc_summary.json", "w"), indent=1)
for k, s in list(summary["kernels"].items())[:18]:
    print(f"{k[:70]:70s} n={s['dispatches']:5d} mfma/busy={s['mfma_busy_over_sq_busy']:.3f} wait={s['wait_any_over_wave']:.2f} stall={s['wait_inst_over_wave']:.2f} ldsconf={s['lds_conflict_over_lds_active']:.2f} fetchKiB={s['fetch_kib_per_dispatch']:.0f} writeKiB={s['write_kib_per_dispatch']:.0f}")
PY
