set -u
export MSH_DEV_KNOBS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for PF in 0 32 16 64 8 32 0; do
  echo "== MSH_ENC_ATT_PF=$PF"
  MSH_ENC_ATT_PF=$PF timeout 300 python tools/enc_attention_microbench.py 2>&1 | grep -E "variant +(0|1|101) " | tail -3
done
timeout 600 python -m pytest tests/test_gpu_enc_attention.py -q 2>&1 | tail -3
bash tools/scratch/ab.sh "MSH_ENC_ATT_PF=0" "MSH_ENC_ATT_PF=32"
