#!/bin/bash
# Build experimental variants of the solve kernels (-DKG_EXP=n) next to the product library: exp/libkarpgpu_exp<n>.so
# Usage: tools/build_exp.sh 1 2 ...   (host objects, device.o and debug_probes.o are reused from the last `make all`)
set -e
cd "$(dirname "$0")/../karpenter_amd/csrc"
mkdir -p ../exp
for n in "$@"; do
  for p in 0 1 2 3 4 5 6 7; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -mllvm -amdgpu-load-store-vectorizer=0 -DKG_EXP=$n -DKG_PART=$p -c dev_part.hip -o ../exp/dev_part${p}_exp$n.o &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libkarpgpu_exp$n.so karpgpu.o catalog.o catalog_ec2.o group.o device.o debug_probes.o ../exp/dev_part?_exp$n.o -lrocprofiler-sdk-roctx
done
