# the row-space tests + the phases of the ill-conditioned large-K fits (scripts/rowspace_large_k.py) on one box
mkdir -p gpurun_out/r06_rs
(timeout 900 python -m pytest tests/test_gpu_rowspace.py tests/test_gpu_condest.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|error" ) > gpurun_out/r06_rs/tests.log 2>&1
FSNAP_ROWSPACE_TIMING=1 timeout 600 python scripts/rowspace_large_k.py 15213 1595 6 > gpurun_out/r06_rs/k1595.log 2>&1
FSNAP_ROWSPACE_TIMING=1 timeout 600 python scripts/rowspace_large_k.py 367900 480 4 > gpurun_out/r06_rs/k480.log 2>&1
tail -4 gpurun_out/r06_rs/tests.log; grep -E "call 2|lstsq on" gpurun_out/r06_rs/k1595.log gpurun_out/r06_rs/k480.log | cut -c1-150
