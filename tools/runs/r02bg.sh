cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bg
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02bg/pytest.log 2>&1; rc=$?; tail -12 gpurun_out/r02bg/pytest.log
( for i in 1 2; do NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py; done ) 2>&1 | grep " N "
exit $rc
