#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes as MI355X_MICROARCH.md prescribes) and MFMA-busy of the streaming
# kernels of BASELINE config 5: one timed step of bench.py --workload streaming.  usage: tools/gpu_pmc_streaming.sh TAG [kernel-regex]
set -u
export MSH_DEV_KNOBS=1
TAG=${1:-spmc}; PAT=${2:-"cross_attention_wide|enc_window_attention|cross_attention_kernel|self_attention_ar"}
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
CMD="python $R/bench.py --workload streaming --steps 1 --warmup 1 --no-stream-profile"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/spmc_${TAG}_$n -o p -- $CMD > /tmp/spmc_${TAG}_$n.log 2>&1)
done
python - "$PAT" <<PY | tee gpurun_out/${TAG}_pmc_streaming_kernels.txt
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
agg = collections.defaultdict(float); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    for f in glob.glob("/tmp/spmc_${TAG}_%s/**/*counter_collection.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
            if not pat.search(k): continue
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("# per dispatch, averaged over every launch of one config-5 step (64 streams, memories growing from 25 to 500 frames); FETCH_SIZE in KiB, doubled (gfx950)")
for k in sorted({k for k, _ in agg}):
    line = "%-44s" % k
    f = agg.get((k, "FETCH_SIZE"), 0) / max(len(cnt.get((k, "FETCH_SIZE"), [1])), 1)
    w = agg.get((k, "WRITE_SIZE"), 0) / max(len(cnt.get((k, "WRITE_SIZE"), [1])), 1)
    mf = agg.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0); gui = agg.get((k, "GRBM_GUI_ACTIVE"), 0) / 8.0
    line += " n=%6d  read %8.2f MB  written %7.2f MB" % (len(cnt.get((k, "FETCH_SIZE"), [])), 2 * f * 1024 / 1e6, w * 1024 / 1e6)
    if gui > 0: line += "  mfma_busy_frac_of_chip %.4f" % (mf / (gui * 1024.0))
    print(line)
PY
