python -m pytest tests/test_gpu_streaming.py -m gpu -q 2>&1 | tail -12
ESL_CHOL_TIMING=1 python scripts/chol_bench.py 2994 2>&1 | grep -v amdgpu | tail -4
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_chol3k -o c3k -- python $GRAFT_REPO_ROOT/scripts/chol_bench.py 2994 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python profiles/summarize_rocpd.py gpurun_out/prof_chol3k/c3k_results.db 2>&1 | head -14
