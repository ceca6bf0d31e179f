#!/bin/bash
# round 4, visit 1: correctness of kernel 1A (fused pack, NB = 9) + reduce 2b, then A/B timings
set -x
O=gpurun_out/r04v1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column_block_shapes or fused_packing or one_wave_triangle or general_k_tiled or tiled_kernel or bit_identical or ridge_fit_k142 or device_solve or mirror or streaming or kernel_variants" > $O/pytest1.log 2>&1
tail -5 $O/pytest1.log
for opt in "" "--option fused_pack=0" "--option reduce=1" "--option mirror_upper=0" "--option fused_pack=0 --option reduce=1 --option mirror_upper=0"; do
  tag=$(echo "$opt" | tr -d ' -' | tr '=' '_'); [ -z "$tag" ] && tag=default
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"], d.get("pipelined",{}).get("ms_per_step"))
PY
done
for shape in "13035 142" "1772880 142" "100000 142" "125000 128" "1772880 110"; do
  set -- $shape
  for opt in "" "--option acc_max_k=128"; do
    [ "$2" != "142" ] && [ -n "$opt" ] && continue
    tag=$1x$2$(echo "$opt" | tr -d ' -' | tr '=' '_')
    timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --rows $1 --cols $2 $opt > $O/bench_$tag.json 2> $O/bench_$tag.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"])
except Exception as e:
    print("$tag FAILED", e)
PY
  done
done
