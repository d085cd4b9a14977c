"""One screen of a gpu_final.sh bundle: python tools/bench_summary.py <tag>  (reads gpurun_out/<tag>_*)."""
import json, sys
tag = sys.argv[1]
print(open(f'gpurun_out/{tag}_pytest.log').read().strip().splitlines()[-1])
d=json.loads(open(f'gpurun_out/{tag}_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'serial', d['serial_steps'])
print('roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline']['traffic'])
enc=0
for k in d['kernels']:
    if k['kernel'].startswith(('enc','conv','group','pack')):
        print(' ', k['kernel'], k['ms_per_launch'], k['frac'], k.get('traffic'), k['total_ms']); enc+=k['total_ms']
print('enc total', round(enc,3))
print('latency', d['latency_ms'])
c=d['c_api_batch']; print('capi', c['value'], c['ms_per_call'], 'vad', c['default_vad']['value'], c['default_vad']['ms_per_call'], 'pcm16', c['default_vad']['pcm16']['value'])
s=d['streaming_config5']; print('stream', s['value'], s['decoder_passes']['us_per_ar_pass'], s['decoder_passes']['us_per_verify_pass'], 'lat1', d['streaming_latency_1stream']['mean_last_update_latency_ms'])
print('step', d['decode_step_us'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_over_cpu'], d['cpu_baseline']['clips_with_ids_equal_to_gpu'])
print('fp8', d['fp8_kv']['value'], 'pcie', d['pcie_inclusive']['value'], '40', d['typical_40_steps']['value'])
t=json.loads(open(f'gpurun_out/{tag}_bench_torchrun1.json').read().strip().splitlines()[-1]); print('torchrun1', t['value'])
p=json.load(open(f'gpurun_out/{tag}_pmc_traffic.json'))
print({g: round(v['traffic_bytes_per_launch']/1e6,1) for g,v in p['groups'].items()})
m=json.load(open(f'gpurun_out/{tag}_pmc_mfma_busy.json'))
for k,v in list(m['kernels'].items()):
    if any(x in k for x in ('mlp_fused','panel_gemm','enc_attention','EpiGnBias','EpiBiasGeluF32','EpiTanh')): print(' ', k[:70], v['mfma_busy_frac_of_chip'])
