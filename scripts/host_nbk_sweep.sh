#!/bin/bash
# host K x K solve against the panel width of the register-blocked Cholesky (FSNAP_CHOL_NBK; default 8 up to 256 columns)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for nbk in 8 12 16 24 32; do echo "NBK=$nbk: $(FSNAP_CHOL_NBK=$nbk python scripts/host_solve_bench.py 128 144 160 192 224)"; done
