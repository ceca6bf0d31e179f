#!/bin/bash
# One GPU visit (every command under its own timeout: a hung command must not eat the GPU budget): GPU test suite, headline bench (single-GPU step and the multi-GPU step in a 1-rank communicator),
# rocprofv3 kernel statistics of the bench.  Usage: gpurun -- 'bash scripts/gpu_visit.sh <tag>'
tag=${1:-visit}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -x --durations=15 > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log
tail -40 $out/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_dist1.json 2> $out/bench_dist1.err; echo "bench dist rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-repack --no-cpu-baseline > $out/bench_norepack.json 2>> $out/bench.err
head -c 1500 $out/bench.json; echo; head -c 600 $out/bench_dist1.json; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err)
ls $out/prof 2>/dev/null | head
