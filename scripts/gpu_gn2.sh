#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GN_ONLY=5
echo "== two-launch kernels (no LDS / no register-resident)"; DDPM_GN_NO_LDS_FWD=1 DDPM_GN_NO_LDS_BWD=1 DDPM_GN_NO_FUSED=1 timeout 300 python scripts/gn_bench.py 2>&1 | tail -6
echo "== same, no dropout"; GN_DROP=0 DDPM_GN_NO_LDS_FWD=1 DDPM_GN_NO_LDS_BWD=1 DDPM_GN_NO_FUSED=1 timeout 300 python scripts/gn_bench.py 2>&1 | tail -6
echo "== same, no dropout no silu"; GN_SILU=0 GN_DROP=0 DDPM_GN_NO_LDS_FWD=1 DDPM_GN_NO_LDS_BWD=1 DDPM_GN_NO_FUSED=1 timeout 300 python scripts/gn_bench.py 2>&1 | tail -6
