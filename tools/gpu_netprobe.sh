#!/bin/bash
# Does the GPU lease reach the network?  ("missing" 1-2: a real checkpoint through the engine and Silero pinned on
# the published model both need downloads.)  Records the answer either way; when a host answers, runs the real-checkpoint
# checks (tools/verify_real_checkpoint.py) and keeps their JSON reports.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
OUT=gpurun_out/${1:-r5}_network_probe.txt
{
  echo "# network probe from the GPU lease, $(date -u +%Y-%m-%dT%H:%M:%SZ)"
  ok=0
  for url in https://huggingface.co https://download.moonshine.ai/model/base-en/quantized/base-en/tokenizer.bin https://raw.githubusercontent.com https://pypi.org; do
    code=$(curl -sI --max-time 8 -o /dev/null -w '%{http_code}' "$url" 2>&1); rc=$?
    echo "curl -sI --max-time 8 $url -> rc=$rc http=$code"
    [ $rc -eq 0 ] && [ "$code" != "000" ] && ok=1
  done
  python - <<'PY'
import socket
for host in ("huggingface.co", "download.moonshine.ai", "github.com"):
    try:
        print("getaddrinfo", host, "->", socket.getaddrinfo(host, 443)[0][4][0])
    except Exception as e:
        print("getaddrinfo", host, "-> FAILED:", e)
PY
  echo "default route: $(ip route 2>/dev/null | head -1 || echo none)"
  if [ $ok -eq 1 ]; then
    echo "network reachable: running tools/verify_real_checkpoint.py"
    timeout 900 python tools/verify_real_checkpoint.py --arch base --hf --report gpurun_out/verify_base_report.json 2>&1 | tail -20
    timeout 600 python tools/verify_real_checkpoint.py --silero --report gpurun_out/verify_silero_report.json 2>&1 | tail -20
    timeout 900 python tools/verify_real_checkpoint.py --ort --report gpurun_out/verify_ort_report.json 2>&1 | tail -20
  else
    echo "VERDICT: no network from the GPU lease -- real checkpoints / the published Silero model cannot be fetched; f1 and f2 stay pinned on what the checkout holds"
  fi
} 2>&1 | tee "$OUT"
