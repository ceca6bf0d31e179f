#!/bin/bash
for i in 1 2; do FSNAP_ROWSPACE_TIMING=1 timeout 600 python scripts/rowspace_large_k.py 15213 1595 6 2>&1 | grep "call\|certified" | cut -c1-90; done
