#!/bin/bash
# usage: tools/experiments/build_variant_x.sh <name> <extra hipcc flags...>   -> scratch/variants/<name>.so
# (the library with r512x_kernel.hip compiled under the extra flags; select it with DSP_AMD_LIB=<path>)
set -e
name=$1; shift
cd /root/repo/dsp_amd/csrc
mkdir -p /root/repo/scratch/variants /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize "$@" -c r512x_kernel.hip -o /tmp/var_$name/r512x_kernel.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/$name.so build/abi_core.o build/abi_stream.o build/abi_multi.o build/abi_unfused.o build/abi_exports.o build/generic_kernels.o build/r512_kernel.o build/r512_fast_rows.o build/r512_fast_kernel.o build/r256_kernel.o build/r4096_kernel.o build/r8192_kernel.o build/wave_fft_kernel.o /tmp/var_$name/r512x_kernel.o
