#!/bin/bash
# scripts/ab_tree.sh : interleaved A/B of the working tree against another CHECKOUT of the repository on one box -- needed when
# the Python side changed with the library (RESDEPTH_HIP_LIB alone cannot swap such builds).  Prepare the other tree with
#   git worktree add -f ablate_libs/old_tree <commit> && (cd ablate_libs/old_tree/resdepth_amd/csrc && bash build.sh)
# (ablate_libs/ is git-ignored but travels with gpurun).  Prints tiles/s, the dominant kernel's TFLOP/s and launch time.
for r in 1 2 3; do
  for T in . ablate_libs/old_tree; do
    (cd $T && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$T', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])")
  done
done
