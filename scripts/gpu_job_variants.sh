for v in "" H1 H2 H3; do
  if [ -z "$v" ]; then unset ESL_HIP_LIB; else export ESL_HIP_LIB=$PWD/object-oriented-slam_amd/csrc/variants/$v.so; fi
  python scripts/map_quick.py C4 2>&1 | grep -v amdgpu.ids
  python scripts/map_quick.py C3 2>&1 | grep -v amdgpu.ids
done
unset ESL_HIP_LIB
ESL_LM_UNFUSED=1 python scripts/map_quick.py C4 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_device_lm.py tests/test_gpu_optimizer.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -25
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_grouped -o grouped -- python $GRAFT_REPO_ROOT/scripts/map_quick.py C4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_grouped | head; python profiles/summarize_rocpd.py gpurun_out/prof_grouped 2>&1 | head -30
