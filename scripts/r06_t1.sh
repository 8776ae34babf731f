mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -4 gpurun_out/smoke.log
timeout 1500 python -m pytest tests/test_gpu_reference_order.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/t1.log 2>&1; tail -15 gpurun_out/t1.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/bench_r06_a.json 2> gpurun_out/bench_r06_a.err; tail -c 600 gpurun_out/bench_r06_a.err; python - <<'PY'
import json
o=json.loads([l for l in open('gpurun_out/bench_r06_a.json') if l.startswith('{')][-1])
print({k:o[k] for k in ('value','ms_per_step','order')})
print('roofline',{k:o['roofline'][k] for k in ('frac','achieved','traffic','kernel_ms','kernel')}, o['roofline'].get('valu'))
print('device_order',{k:(v if not isinstance(v,dict) else v.get('solves_per_s')) for k,v in o.get('device_order',{}).items()})
print('cpu',o.get('cpu_baseline',{}).get('value'),o.get('cpu_baseline',{}).get('kind'))
print('parity ref',o.get('parity',{}).get('reference_order'))
print('isolated',o.get('isolated')); print('errors',o.get('side_run_errors'))
print('gear',o.get('gear_shift_4096_reference_order')); print('mov',o.get('moving_obstacles_1024',{}).get('reference_order'))
PY
