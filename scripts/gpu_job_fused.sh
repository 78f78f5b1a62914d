set -x
python -m pytest tests/test_gpu_device_lm.py tests/test_gpu_optimizer.py tests/test_gpu_sharded.py "tests/test_gpu_fullsize.py" -m gpu -q -x 2>&1 | tail -40
python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>gpurun_out/bench_fused.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('FUSED value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['kernel_ms'], 'roof', d['roofline']['avg_launch_ms'], d['config']['lm_iterations_per_step'], d['config']['lm_trials_per_step'])"
ESL_LM_UNFUSED=1 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>>gpurun_out/bench_fused.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('UNFUSED value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['kernel_ms'], 'roof', d['roofline']['avg_launch_ms'])"
tail -5 gpurun_out/bench_fused.err
