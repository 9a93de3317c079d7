mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv1d_d4_gpu.py -m gpu -x -q 2>&1 | tail -5
for d in 0 1; do
  SVC_CONV_DIRECT4=$d python scripts/bench_front_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r11k_front_conv_d4_$d.txt
  SVC_CONV_DIRECT4=$d timeout 300 python bench.py --mode infer --no-extras --no-cpu-baseline --no-pmc --no-host-io > gpurun_out/r11k_infer_d4_$d.json 2> gpurun_out/r11k_infer_d4_$d.err
  SVC_CONV_DIRECT4=$d timeout 300 python bench.py --mode infer --half --no-extras --no-cpu-baseline --no-pmc --no-host-io --no-roofline > gpurun_out/r11k_half_d4_$d.json 2> gpurun_out/r11k_half_d4_$d.err
done
cat gpurun_out/r11k_front_conv_d4_*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r11k_*.json')):
    for line in open(f):
        if line.startswith('{'):
            d=json.loads(line); print(f, d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))
PY
