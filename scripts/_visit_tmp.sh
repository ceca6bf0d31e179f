mkdir -p gpurun_out/v26; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --rows 1000000 --cols 31 --steps 30 --warmup 3 > gpurun_out/v26/bench_1000000x31.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --rows 15213 --cols 31 --steps 30 --warmup 3 > gpurun_out/v26/bench_15213x31.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --rows 10000000 --cols 31 --steps 30 --warmup 3 > gpurun_out/v26/bench_10000000x31.json 2>/dev/null
python -c "
import json
for s in ('1000000x31','15213x31','10000000x31'):
    d=json.loads(open('gpurun_out/v26/bench_%s.json'%s).read()); print(s, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['weighting_kernel']['frac'])"
