python -m pytest tests/test_harness.py tests/test_gpu_device_lm.py tests/test_gpu_fullsize.py::test_c4_full_graph_final_states_match_oracle -m gpu -q -s 2>&1 | grep -v "^$" | tail -25
( time python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err ) 2>&1 | tail -4
tail -3 gpurun_out/r2_bench_default.err
python -c "
import json
d = json.load(open('gpurun_out/r2_bench_default.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('achieved','frac','avg_launch_ms')})
print('repeat', d.get('repeat_blocks'))
print('host', d.get('host'))
for k, v in d.get('slam', {}).items(): print('slam', k, v['value'], v['ms_per_optimize'], v['roofline']['achieved'], v['roofline']['frac'], v.get('cpu_baseline', {}).get('value'), v.get('speedup_vs_cpu_port'))
for k, v in d.get('fit', {}).items(): print('fit', k, v['ms_per_frame_kernel'], v['ms_per_frame_host_call'], v['roofline']['achieved'], v.get('cpu_port_ms_per_frame'))
print('stream', d.get('streaming_c5'))
print('cpu', d['cpu_baseline']['value'], d['speedup_vs_cpu_port'])
"
