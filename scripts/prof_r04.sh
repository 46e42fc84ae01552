#!/bin/bash
# GPU box: the round's rocprofv3 evidence, one configuration per call (keeps each gpurun call short).  Usage: scripts/prof_r04.sh WHICH
set -o pipefail
case "$1" in
  L6) bash scripts/prof_round.sh r04_L6 --steps 3 --warmup 1 ;;
  L1) bash scripts/prof_round.sh r04_L1 --level 1 --steps 2 --warmup 1 ;;
  L9) bash scripts/prof_round.sh r04_L9 --level 9 --steps 1 --warmup 1 ;;
  inflate) bash scripts/prof_round.sh r04_inflate --op inflate --steps 3 --warmup 1 ;;
  cont) bash scripts/prof_round.sh r04_cont_L6 --continuous --steps 2 --warmup 1 ;;
  cont1) bash scripts/prof_round.sh r04_cont_L1 --continuous --level 1 --steps 1 --warmup 1 ;;
esac
