mkdir -p gpurun_out/v28; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/v28/gpu_tests.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/v28/gpu_tests.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python examples/transpose_trick_stream.py 2>&1 | tail -1
python examples/multi_gpu_fit.py 2>&1 | tail -2
