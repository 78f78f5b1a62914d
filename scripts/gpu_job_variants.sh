python scripts/map_quick.py C4 2>&1 | grep -v amdgpu.ids
python scripts/map_quick.py C3 2>&1 | grep -v amdgpu.ids
ESL_LM_STEP_OLD=1 python scripts/map_quick.py C4 2>&1 | grep -v amdgpu.ids
ESL_LM_STEP_OLD=1 python scripts/map_quick.py C3 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_device_lm.py tests/test_gpu_optimizer.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -25
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rows -o rows -- python $GRAFT_REPO_ROOT/scripts/map_quick.py C4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python profiles/summarize_rocpd.py gpurun_out/prof_rows/rows_results.db 2>&1 | head -30
