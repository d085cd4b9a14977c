#!/bin/bash
# PMC session: wide-tile A/B, then rocprofv3 counter passes (MFMA busy, LDS conflicts, HBM bytes) per kernel.
set -u
TAG=${1:-pmc}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
for wide in 96 100000; do
  echo "== bench MSH_DEC_WIDE_M=$wide"
  MSH_DEC_WIDE_M=$wide timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_w${wide}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_w${wide}.json"))
print("value",d["value"],"ms_per_step",d["ms_per_step"],"latency",d["latency_ms"]["p50_total"])
for r in d["kernels"]:
    if r["kernel"].startswith("dec_"): print(f"  {r['kernel']:24s} ms/launch={r['ms_per_launch']:.4f} total={r['total_ms']:.2f}")
PY
done
echo "== counters available"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy)\b" | sort -u | tr '\n' ' ' | cut -c1-3000 > gpurun_out/${TAG}_counter_names.txt
wc -c gpurun_out/${TAG}_counter_names.txt
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency"
pass() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- $CMD > /tmp/pmc_${TAG}_$name.log 2>&1)
  tail -1 /tmp/pmc_${TAG}_$name.log | cut -c1-200
  ls /tmp/pmc_${TAG}_$name 2>/dev/null | head -5
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python - <<PY
import csv,glob,collections,json
out={}
for name in ("sq","fetch","write"):
    files=glob.glob(f"/tmp/pmc_${TAG}_{name}/*counter_collection.csv")
    if not files: print("no counter file for",name); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(files[0])):
        k=r["Kernel_Name"].replace("msh::(anonymous namespace)::","")[:110]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        key=(r["Dispatch_Id"],)
        if key not in seen:
            seen.add(key); cnt[k]+=1
    for k,v in agg.items():
        out.setdefault(k,{"dispatches":cnt[k]}).update(v)
json.dump(out,open("gpurun_out/${TAG}_pmc_summary.json","w"),indent=1)
for k,v in sorted(out.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:16]:
    d=v["dispatches"]
    mf=v.get("SQ_VALU_MFMA_BUSY_CYCLES",0); busy=v.get("SQ_BUSY_CYCLES",1)
    print(f"{k[:86]:86s} n={d:4d} mfma_busy/sq_busy={mf/max(busy,1):.3f} ldsconf/ldsact={v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} wait_inst/wave={v.get('SQ_WAIT_INST_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f} wait_any/wave={v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f} fetchKB/disp={v.get('FETCH_SIZE',0)/d:.0f} writeKB/disp={v.get('WRITE_SIZE',0)/d:.0f}")
PY
