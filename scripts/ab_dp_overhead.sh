#!/bin/bash
# Overhead of the data-parallel machinery at world size 1 (RCCL process group, loss normaliser, bucketed all-reduce launched from the
# weight-gradient stream), interleaved on one box: plain | --force-dist with 16 MB buckets | 64 MB (one bucket) | 4 MB
for r in 1 2; do
  for v in "plain" "--force-dist" "--force-dist --bucket-mb 64" "--force-dist --bucket-mb 4"; do
    a="$v"; [ "$v" = plain ] && a=""
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --no-prof $a 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'round $r', d['value'], d['step_ms_median'])"
  done
done
