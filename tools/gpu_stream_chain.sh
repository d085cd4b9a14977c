#!/bin/bash
# What every kernel group of a streaming auto-regressive step costs INSIDE its replayed graph (config 5, 64 streams): the step
# with one group left in (MSH_STREAM_AR_MASK), timed by decode_full's own HIP events.  Tokens are garbage under a partial mask.
set -u
export MSH_DEV_KNOBS=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
TAG=${1:-chain}
: > gpurun_out/${TAG}_stream_chain_costs.txt
for M in ${MASKS:-0x3ff 0x1 0x2 0x4 0x8 0x10 0x20 0x40 0x80 0x100 0x0 0x3 0xff}; do
  MSH_STREAM_AR_MASK=$M timeout 600 python bench.py --workload streaming --steps 1 --warmup 1 --no-stream-profile > gpurun_out/${TAG}_m.json 2> gpurun_out/${TAG}_m.err
  python - <<PY | tee -a gpurun_out/${TAG}_stream_chain_costs.txt
import json
d = json.loads(open("gpurun_out/${TAG}_m.json").read().strip().splitlines()[-1])
p = d["streaming"]["decoder_passes"]
names = ["ln_qkv", "self_attention", "o_proj", "ln_cross_q", "cross_attention", "cross_o", "ln_fc1_swiglu", "fc2", "final_ln_lm_head"]
m = int("$M", 16)
what = "whole step" if m == 0x3ff else ("advance only" if m == 0 else " + ".join(n for i, n in enumerate(names) if (m >> i) & 1))
layers = 12
per = p["us_per_ar_pass"]
print(f"mask {m:#05x} {what:60s} {per:8.1f} us per pass  ({p['ar_passes_per_step']:.0f} passes)")
PY
done
