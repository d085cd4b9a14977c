#!/bin/bash
# r3u: C-API tests + batch-call timing, then the streaming workload under rocprofv3 with a gap analysis of the kernel trace
set -u
tag=${1:-r3u}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_capi.py tests/test_gpu_capi_threads.py -q -x 2>&1 | tail -2
MSH_HOST_TIMING=1 timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-fp8 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c_api", d["config"].get("c_api_batch_value"), "default_vad", d["config"].get("c_api_batch_default_vad_value"))
PY
grep "batch call" gpurun_out/${tag}_bench.err | tail -12
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_prof -o ${tag} -- python $R/bench.py --workload streaming --steps 1 --warmup 1 --no-stream-profile > $R/gpurun_out/${tag}_sbench.json 2> $R/gpurun_out/${tag}_sbench.err
cd $R
cut -c1-300 gpurun_out/${tag}_sbench.json
f=$(find /tmp/${tag}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${tag}_stream_kernel_stats.csv
t=$(find /tmp/${tag}_prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, re, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"msh::\(anonymous namespace\)::","",r["Kernel_Name"]).split("(")[0]
    n=re.sub(r"^void ","",n)
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n[:70]))
rows.sort()
# the last 40 % of the trace = the timed step; gaps between consecutive kernels
lo=int(len(rows)*0.55)
rows=rows[lo:]
dur=collections.defaultdict(float); cnt=collections.Counter(); gap_after=collections.defaultdict(float)
busy=0; gaps=0
for i,(s,e,n) in enumerate(rows):
    dur[n]+=e-s; cnt[n]+=1; busy+=e-s
    if i+1<len(rows):
        g=max(0,rows[i+1][0]-e)
        if g<200000: gap_after[n]+=g; gaps+=g
print("kernels %d busy %.1f ms gaps(<200us) %.1f ms span %.1f ms"%(len(rows),busy/1e6,gaps/1e6,(rows[-1][1]-rows[0][0])/1e6))
for n,_ in sorted(dur.items(),key=lambda x:-x[1]-gap_after[x[0]])[:32]:
    print("%-70s n=%6d avg=%7.2f us gap_after=%6.2f us total=%7.1f ms"%(n,cnt[n],dur[n]/cnt[n]/1e3,gap_after[n]/cnt[n]/1e3,(dur[n]+gap_after[n])/1e6))
PY
