#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file: bash tools/experiments/res_usage.sh dsp_amd/csrc/r4096_kernel.hip [-D...]
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /LDS Size/ {print n, "vgpr", v, "agpr", a, "scratch", s, "occ", o}'
