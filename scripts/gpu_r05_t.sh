#!/bin/bash
mkdir -p gpurun_out/r05_t
timeout 600 python scripts/upload_probe.py > gpurun_out/r05_t/upload_probe.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_t/upload_probe.txt
