#!/bin/bash
# rocprofv3 PMC passes over the encoder-attention microbenchmark (development library): SQ counters of one variant.
# usage: tools/gpu_pmc_attn.sh <tag> <variant>
set -u
TAG=$1; VAR=${2:-100}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp MSH_DEV_KNOBS=1
cat > /tmp/attn_one.py <<PY
import sys
sys.path.insert(0, "$R")
from moonshine_amd.hip_api import load_dev_library
lib = load_dev_library()
print(lib.msh_test_enc_attention($VAR, 256, 415, 416, 8, 3, None))
PY
pass_() {
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmca_${TAG}_${name} -o p -- python /tmp/attn_one.py > /tmp/pmca_${TAG}_${name}.log 2>&1)
  python - "$name" <<PY
import csv, glob, collections, sys
files = glob.glob("/tmp/pmca_${TAG}_" + sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not files:
    print(open("/tmp/pmca_${TAG}_" + sys.argv[1] + ".log").read()[-1500:]); sys.exit(0)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(files[0])):
    if "enc_attention" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in sorted(agg): print("%-34s %18.0f per dispatch (%d dispatches)" % (c, agg[c] / n[c], n[c]))
PY
}
{
echo "variant $VAR"
pass_ a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass_ b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
pass_ c SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE
} > gpurun_out/${TAG}_pmc_attn_v${VAR}.txt 2>&1
cat gpurun_out/${TAG}_pmc_attn_v${VAR}.txt
