#!/bin/bash
# Builds, on the CPU, the library with AddressSanitizer + UBSan on its HOST code (device code as usual) and the driver
# that calls the host-pointer entry points, into scratch/san/.  tools/san/run.sh runs the driver on a GPU box.
set -e
R=$(cd $(dirname $0)/../.. && pwd)
O=$R/scratch/san
HIPCC=/opt/rocm/bin/hipcc
CLANG=/opt/rocm/lib/llvm/bin/clang++
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer"
mkdir -p $O
for f in abi_core abi_stream abi_multi abi_unfused abi_exports generic_kernels r512_kernel r512_fast_kernel r512_fast_rows r512x_kernel r256_kernel r4096_kernel r8192_kernel wave_fft_kernel; do
  ( $HIPCC -O3 -g1 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize $SAN -c $R/dsp_amd/csrc/$f.hip -o $O/$f.o ) &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=address,undefined -shared-libsan -o $O/libdsp_stft_hip.so $O/*.o
$CLANG -O1 -g -std=c++17 -fno-gpu-sanitize -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -I$R/include $R/tools/san/host_paths.cpp -o $O/host_paths -L$O -ldsp_stft_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,$(dirname $($CLANG -print-file-name=libclang_rt.asan-x86_64.so))
ls -la $O/libdsp_stft_hip.so $O/host_paths
