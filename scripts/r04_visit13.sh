#!/bin/bash
# round 4, visit 13: kernel 13A with the rows of R requested a block ahead (A/B against the round-3 staging), the two new
# parity tests, the row-space test file
O=gpurun_out/r04v13; mkdir -p $O
(for s in "1000000 128" "1000000 110" "1000000 96" "1000000 64" "1000000 31" "100003 128" "6001 100"; do
   for f in 0 1; do
     echo "new: $(timeout 120 tools/trsm_check $s $f 5 | head -1)"
     echo "old: $(FSNAP_TRSM_ACC=1 timeout 120 tools/trsm_check $s $f 5 | head -1)"
   done
 done) > $O/trsm_ab.txt 2>&1
cat $O/trsm_ab.txt
timeout 900 python -m pytest tests/test_gpu_rowspace.py -x -q > $O/rowspace.log 2>&1; tail -3 $O/rowspace.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bound_in_caller or falls_back" > $O/newtests.log 2>&1; tail -5 $O/newtests.log
