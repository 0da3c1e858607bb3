#!/bin/bash
# Build experimental variants of the device code (-DKG_EXP=n) next to the product library: exp/libkarpgpu_exp<n>.so
# Usage: tools/build_exp.sh 1 2 ...   (objects of the host side are reused from the last `make all`)
set -e
cd "$(dirname "$0")/../karpenter_amd/csrc"
mkdir -p ../exp
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DKG_EXP=$n -c device.hip -o ../exp/device_exp$n.o &
done
wait
for n in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libkarpgpu_exp$n.so karpgpu.o catalog.o catalog_ec2.o ../exp/device_exp$n.o
done
