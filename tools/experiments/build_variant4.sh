#!/bin/bash
# usage: scratch/build_variant4.sh <name> <extra hipcc flags...>   -> scratch/variants/<name>.so (4096 kernel variants)
set -e
name=$1; shift
cd /root/repo/dsp_amd/csrc
mkdir -p /root/repo/scratch/variants /tmp/var_$name
for f in abi_core abi_stream abi_multi abi_unfused abi_exports generic_kernels r512_kernel r512_fast_kernel r512x_kernel r256_kernel r8192_kernel wave_fft_kernel; do
  [ -f build/$f.o ] || make -s build/$f.o
done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize "$@" -c r4096_kernel.hip -o /tmp/var_$name/r4096_kernel.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|Occupancy" | paste - - - - | grep "ILb1ELb0ELb0" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[^ ]*hip:[0-9:]* remark://g' | cut -c1-200
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/$name.so build/abi_core.o build/abi_stream.o build/abi_multi.o build/abi_unfused.o build/abi_exports.o build/generic_kernels.o build/r512_kernel.o build/r512_fast_rows.o build/r512_fast_kernel.o build/r512x_kernel.o build/r256_kernel.o build/r8192_kernel.o build/wave_fft_kernel.o /tmp/var_$name/r4096_kernel.o
