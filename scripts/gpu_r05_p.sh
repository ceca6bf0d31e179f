#!/bin/bash
mkdir -p gpurun_out/r05_p
FSNAP_PROFILE_CALL=1 timeout 900 python scripts/rowspace_large_k.py > gpurun_out/r05_p/profile.txt 2>&1; grep -v "^\[fsnap" gpurun_out/r05_p/profile.txt | head -90
