#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2e
O=$PWD/gpurun_out/2e
show() { python - $1 "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2],'value %.4g'%d['value'],'step ms %.4f'%d['ms_per_step'],'kernel %.4f'%d['roofline']['kernel_ms_avg'],'red %.4f'%d['roofline']['reduce_kernel_ms_avg'],'launch %.4f sync %.4f solve %.4f'%(d['step_host_launch_ms_avg'],d['step_wait_gpu_ms_avg'],d['step_host_solve_ms_avg']))
PY
}
timeout 200 python bench.py --no-cpu-baseline > $O/single.json 2>> $O/err; show $O/single.json single
timeout 200 python bench.py --no-cpu-baseline --force-dist > $O/dist.json 2>> $O/err; show $O/dist.json force-dist
timeout 200 python bench.py --no-cpu-baseline --force-dist > $O/dist2.json 2>> $O/err; show $O/dist2.json force-dist
tail -5 $O/err
