#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2f
O=$PWD/gpurun_out/2f
timeout 200 python bench.py --no-cpu-baseline --force-dist > $O/dist.json 2> $O/err
echo "stdout lines: $(wc -l < $O/dist.json)"; head -c 300 $O/dist.json; echo
grep -c "RCCL version" $O/err
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/tr.json 2> $O/err2
echo "torchrun stdout lines: $(wc -l < $O/tr.json)"; head -c 200 $O/tr.json; echo
