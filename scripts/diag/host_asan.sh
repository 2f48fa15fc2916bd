#!/bin/bash
# On the GPU box (scratch snapshot): rebuild liblpipm.so with AddressSanitizer + UBSan on the HOST code only (GPU code
# objects are untouched: GPU sanitizers are not available on this pool) and run the host-logic exerciser under it.
set -e
mkdir -p gpurun_out
ASAN_DIR=$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so))
make -C lp_amd/csrc clean > /dev/null
make -C lp_amd/csrc -j16 CXXFLAGS="-O1 -g -std=c++20 -fPIC -pthread --offload-arch=gfx950 -Wall -Wno-unused-function -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer" OUT=../lib/liblpipm_unsanitized_link.so > gpurun_out/asan_make.log 2>&1 || true
ls lp_amd/lib/obj/*.o > /dev/null
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -shared-libsan -o lp_amd/lib/liblpipm.so lp_amd/lib/obj/*.o
/opt/rocm/lib/llvm/bin/clang++ -std=c++20 -O1 -g -fsanitize=address -fsanitize=undefined -shared-libsan -I include scripts/diag/host_asan_driver.cpp -o /tmp/host_asan_driver -L lp_amd/lib -llpipm -Wl,-rpath,$PWD/lp_amd/lib -Wl,-rpath,$ASAN_DIR
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout -k 10 300 /tmp/host_asan_driver
