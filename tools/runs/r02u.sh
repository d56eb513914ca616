cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02u
timeout -k 10 600 python -m pytest tests/test_debug_bvh.py tests/test_scene_io.py -m gpu -x -q > gpurun_out/r02u/pytest.log 2>&1; tail -8 gpurun_out/r02u/pytest.log
