#!/bin/bash
# rocprofv3 PMC pass over ONE bench step (no latency / cpu / streaming legs); prints the counters of kernels matching $2.
# usage: tools/gpu_pmc_kernel.sh <tag> <kernel-regex> <counter> [<counter> ...]   (max 8 SQ counters per pass)
set -u
TAG=$1; PAT=$2; shift 2
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api"
(cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmck_${TAG} -o p -- $CMD > /tmp/pmck_${TAG}.log 2>&1)
python - "$PAT" <<PY
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
files = glob.glob("/tmp/pmck_${TAG}/**/*counter_collection.csv", recursive=True)
if not files:
    print(open("/tmp/pmck_${TAG}.log").read()[-2000:]); sys.exit(1)
agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(files[0])):
    k = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]
    if not pat.search(k): continue
    agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
for k in cnt:
    print(k, "dispatches", cnt[k])
    for (kk, c), v in sorted(agg.items()):
        if kk == k: print("   %-32s %16.0f per dispatch" % (c, v / cnt[k]))
PY
