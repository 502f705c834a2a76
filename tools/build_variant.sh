#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> tools/bin/libcmvae_<name>.so: the library with other
# compile-time settings (-DCMVAE_ABLATE=<n>, -DCMVAE_BF3_MIN_ROWS=<rows>), for CMVAE_HIP_LIB=<path> comparison runs
set -e
cd "$(dirname "$0")/../cmvae_amd/csrc"
name=$1; shift
mkdir -p ../../tools/bin/obj_$name
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $f -o ../../tools/bin/obj_$name/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 ../../tools/bin/obj_$name/*.o -o ../../tools/bin/libcmvae_$name.so
echo built tools/bin/libcmvae_$name.so
