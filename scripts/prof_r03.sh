#!/bin/bash
# GPU box: the round's rocprofv3 evidence, one configuration per call (keeps each gpurun call short).  Usage: scripts/prof_r03.sh WHICH
set -o pipefail
case "$1" in
  L6) bash scripts/prof_round.sh r03_L6 --steps 3 --warmup 1 ;;
  L1) bash scripts/prof_round.sh r03_L1 --level 1 --steps 2 --warmup 1 ;;
  L9) bash scripts/prof_round.sh r03_L9 --level 9 --steps 1 --warmup 1 ;;
  inflate) bash scripts/prof_round.sh r03_inflate --op inflate --steps 3 --warmup 1 ;;
  L1fw) bash scripts/prof_round.sh r03_L1_fastwin_1gib --level 1 --gib 1 --lz fastwin --steps 3 --warmup 1 ;;
esac
