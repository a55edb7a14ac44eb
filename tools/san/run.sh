#!/bin/bash
# Runs the host-sanitizer driver built by tools/san/build.sh (an executable linked against the sanitizer runtime: nothing
# is preloaded).  HIP's runtime maps memory where the address sanitizer expects its shadow gap to stay untouched; leak
# checking would list the runtime's own process-lifetime allocations.
R=$(cd $(dirname $0)/../.. && pwd)
export ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1
$R/scratch/san/host_paths
