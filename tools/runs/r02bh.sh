cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "stack_bound or c5" 2>&1 | grep -v amdgpu | grep "quad_max_stack\|passed\|failed\|Error" 
