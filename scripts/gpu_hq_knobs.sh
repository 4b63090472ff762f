#!/bin/bash
# same-box A/B of the round-5 side-stream defaults on BASELINE config 5's per-GPU work (CelebA-HQ 256x256, B = 2, full Trainer.step)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" python scripts/hq_step.py 40 train 2>/dev/null | tail -1 | sed "s/^/$* : /"; }
for rep in 1 2 3; do
  run DDPM_WGRAD1_CUS=128
  run DDPM_WGRAD1_CUS=256
  run DDPM_WGRAD1_MIN_P=16384
  run DDPM_WGRAD3_TAIL_BLOCKS=0
  run DDPM_SLAB_FLUSH_ROWS=6
done
