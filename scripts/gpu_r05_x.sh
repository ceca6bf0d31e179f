#!/bin/bash
mkdir -p gpurun_out/r05_x
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"; echo "nproc: $(nproc)"; } > gpurun_out/r05_x/quota.txt; cat gpurun_out/r05_x/quota.txt
FSNAP_ROWSPACE_TIMING=1 timeout 600 python scripts/rowspace_large_k.py 15213 1595 6 2>&1 | grep "call 2\|certified\|prepare (all)\|product" | tail -5 | cut -c1-100
