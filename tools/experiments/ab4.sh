#!/bin/bash
# usage: tools/experiments/ab4.sh v1 v2 ... : times the 4096 workload for each variant, twice round-robin
for rep in 1 2; do
for v in "$@"; do
  bash tools/experiments/run_env4.sh $v "X=1"
done
done
