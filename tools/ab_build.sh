#!/bin/bash
# build an A/B variant of libsdt_gpu.so into gpurun_ab/: usage: ab_build.sh <name> "<extra hipcc flags>"
# (only sdt_gpu.hip is recompiled with the flags; the other objects are the in-tree ones -- run `make` first)
set -e
cd "$(dirname "$0")/../soapdenovo-trans_amd/csrc"
mkdir -p ../../gpurun_ab
NAME=$1; FLAGS=$2
hipcc $FLAGS -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -c -o /tmp/sdt_gpu_$NAME.o sdt_gpu.hip
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -o ../../gpurun_ab/libsdt_gpu_$NAME.so /tmp/sdt_gpu_$NAME.o sdt_gpu_graph.o sdt_mem.o sdt_scatter_seq_a.o sdt_scatter_seq_b.o sdt_scatter_seq_c.o sdt_scatter_seq_d.o
echo built gpurun_ab/libsdt_gpu_$NAME.so
