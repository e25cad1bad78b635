#!/bin/bash
# A/B variant whose flags reach every translation unit (bucket geometry): all .hip files are recompiled.  usage: ab_build_all.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../soapdenovo-trans_amd/csrc"
mkdir -p ../../gpurun_ab /tmp/ab_$1
NAME=$1; FLAGS=$2
for f in sdt_gpu sdt_pipeline sdt_sharded sdt_pass2 sdt_mapstage sdt_gpu_graph sdt_mem sdt_scatter_seq_a sdt_scatter_seq_b sdt_scatter_seq_c sdt_scatter_seq_d; do
  hipcc -DSDT_TUNING $FLAGS -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -c -o /tmp/ab_$NAME/$f.o $f.hip &
done
wait
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -o ../../gpurun_ab/libsdt_gpu_$NAME.so /tmp/ab_$NAME/*.o
echo built gpurun_ab/libsdt_gpu_$NAME.so
