#!/bin/bash
# session 3: parity, then group-count / gemm-mode A-B at batch 256, batch-1 profile, rocprofv3 kernel stats
set -u
TAG=${1:-ab2}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("value",d["value"],"ms_per_step",d["ms_per_step"],"latency",d.get("latency_ms"))
    for r in d["kernels"]: print(f"  {r['kernel']:24s} {r['bound']:5s} {r['achieved']:9.1f} {r['unit']:8s} frac={r['frac']:.3f} ms/launch={r['ms_per_launch']:.4f} total={r['total_ms']:.2f}")
except Exception as e:
    print("bench failed", e)
PY
}
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/${TAG}_pytest.log
for cfg in "2 1" "2 2" "2 4" "2 8" "3 4"; do
  set -- $cfg
  echo "== bench MSH_GEMM_MODE=$1 MSH_DEC_GROUPS=$2"
  MSH_GEMM_MODE=$1 MSH_DEC_GROUPS=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/${TAG}_bench_m$1_g$2.json 2> gpurun_out/${TAG}_bench_m$1_g$2.err
  show gpurun_out/${TAG}_bench_m$1_g$2.json | head -3
  tail -2 gpurun_out/${TAG}_bench_m$1_g$2.err
done
echo "== mode 3 kernels"; show gpurun_out/${TAG}_bench_m3_g4.json | grep -E "enc_|conv|cross_kv"
echo "== batch 1 profile"
timeout 600 python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_b1.json 2> gpurun_out/${TAG}_bench_b1.err
show gpurun_out/${TAG}_bench_b1.json
echo "== rocprofv3 kernel stats (batch 256, default config)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o prof -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /tmp/prof_${TAG}.log 2>&1)
tail -2 /tmp/prof_${TAG}.log | cut -c1-300
mkdir -p gpurun_out/${TAG}_prof
find /tmp/prof_${TAG} -name "*stats*.csv" -exec cp {} gpurun_out/${TAG}_prof/ \; 2>/dev/null
find /tmp/prof_${TAG} -type f | head; 
python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/${TAG}_prof/*kernel_stats*.csv"):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows))
    for r in rows[:28]:
        print("  %-90s calls=%s total_ms=%.3f avg_us=%.2f pct=%s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
