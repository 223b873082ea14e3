#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]  -> variants/librtamd_<name>.so (for tools/ab.sh)
set -e
name=$1; shift
mkdir -p variants
cd raytracer-rust_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-unused-result "$@" -shared -o ../../variants/librtamd_$name.so rt_kernels.hip rt_api.hip bvh_gpu.hip rt_host.cpp bvh.cpp -L/opt/rocm/lib -lrccl -lz -lpthread
