#!/bin/bash
mkdir -p gpurun_out/r05_s
timeout 900 python scripts/class_fit_survey.py > gpurun_out/r05_s/class_fit_survey.txt 2>&1; cat gpurun_out/r05_s/class_fit_survey.txt | grep -v amdgpu.ids
