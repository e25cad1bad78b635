#!/bin/bash
# build an A/B variant of libsdt_gpu.so into gpurun_ab/: usage: ab_build.sh <name> "<extra hipcc flags>" [unit]
# (only ONE translation unit -- sdt_pipeline.hip unless named -- is recompiled with the flags; the other objects are the in-tree ones: run `make` first)
set -e
cd "$(dirname "$0")/../soapdenovo-trans_amd/csrc"
mkdir -p ../../gpurun_ab
NAME=$1; FLAGS=$2; UNIT=${3:-sdt_pipeline}
hipcc -DSDT_TUNING $FLAGS -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -c -o /tmp/ab_${UNIT}_$NAME.o $UNIT.hip
OTHERS=$(for o in sdt_gpu sdt_pipeline sdt_sharded sdt_pass2 sdt_mapstage sdt_gpu_graph sdt_mem sdt_scatter_seq_a sdt_scatter_seq_b sdt_scatter_seq_c sdt_scatter_seq_d; do [ $o != $UNIT ] && echo $o.o; done)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -o ../../gpurun_ab/libsdt_gpu_$NAME.so /tmp/ab_${UNIT}_$NAME.o $OTHERS
echo built gpurun_ab/libsdt_gpu_$NAME.so
