#!/bin/bash
# End-to-end A/B of the dispatch switches on one box: alternate baseline / toggle, train p50 (ms/step) and sampling ms/step.
run() {
  t=$(env $1 python scripts/step_jitter.py 30 2>/dev/null | head -1 | sed 's/per-step ms: //')
  if [ -z "$TRAIN_ONLY" ]; then s=$(env $1 python scripts/sample_only.py 60 2>/dev/null | tail -1); fi
  echo "$1 | train $t | sample $s"
}
for rep in 1 2; do
  run "BASE=1"
  for t in ${TOGGLES:-DDPM_GN_NO_FUSED=1 DDPM_FUSED_ATTENTION=0 DDPM_NO_XCD_SWIZZLE=1 DDPM_CONV_HALO_MINC=256 DDPM_GEMM_NO_T64=1 DDPM_SIDE_STREAM=0}; do run "$t"; done
done
