#!/bin/bash
# builds kernarg_probe (run it on a GPU box): three translation units, the third with kernarg preloading
set -e
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc; F="--offload-arch=gfx950 -O3 -std=c++17"
$H $F -c k_struct.hip -o k_struct.o
$H $F -c k_plain.hip -o k_plain.o
$H $F -DKNAME=step_preload -DLNAME=launch_preload -mllvm -amdgpu-kernarg-preload-count=8 -c k_plain.hip -o k_preload.o
$H $F -c main.cpp -o main.o
$H --offload-arch=gfx950 main.o k_struct.o k_plain.o k_preload.o -o kernarg_probe
