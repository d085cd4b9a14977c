#!/bin/bash
# round 4: why the serial steps slow down with the two-stage query kernel although the kernel itself is faster
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TAG=${1:-r4v}
FLAGS="--steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-streaming --no-pcie --no-typical --no-c-api --no-fp8"
timeout 300 python bench.py $FLAGS > gpurun_out/${TAG}_two_stage.json 2>/dev/null
MSH_XATTN_QT=1 timeout 300 python bench.py $FLAGS > gpurun_out/${TAG}_merged.json 2>/dev/null
timeout 300 python bench.py --in-flight 1 $FLAGS > gpurun_out/${TAG}_two_stage_serial.json 2>/dev/null
MSH_XATTN_QT=1 timeout 300 python bench.py --in-flight 1 $FLAGS > gpurun_out/${TAG}_merged_serial.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4v_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], 'serial', d['config'].get('serial_steps_value'), 'profiled_step_ms', d.get('profiled_step_ms'))
    print('   ', {k['kernel']: k['ms_per_launch'] for k in d['kernels']})
PY
