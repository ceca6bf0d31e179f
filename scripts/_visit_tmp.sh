mkdir -p gpurun_out/v20; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for ring in 0 1 3; do for shp in "367900 480 100" "15213 1595 100" "13035 142 150" "200000 1000 50"; do set -- $shp
timeout 120 python bench.py --rows $1 --cols $2 --steps 20 --warmup 3 --preheat $3 --no-cpu-baseline --scaling strong --option tiled_ring=$ring 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ring $ring', '$1 x $2', round(r['kernel_ms_avg']*1e3,1),'us', round(r['frac'],3), 'step', round(d['ms_per_step']*1e3,1))"
done; done; done 2>&1 | tee gpurun_out/v20/ring_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_assembly.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
