#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run27
mkdir -p $O
python tools/gen_fastq.py /tmp/r30 30000000 150 20000 > /dev/null 2>&1
cp /tmp/r30/lib.cfg /tmp/r30.cfg
cat /tmp/r30/reads.fq > /dev/null
for i in 1 2; do
( time soapdenovo-trans_amd/csrc/sdt-pregraph pregraph -s /tmp/r30.cfg -K 31 -p 16 -o /tmp/o1 --hash-only ) 2>&1 | grep -E "sdt-pregraph\]|real"
done
( time soapdenovo-trans_amd/csrc/sdt-pregraph pregraph -s /tmp/r30.cfg -K 31 -p 16 -o /tmp/o2 ) 2>&1 | grep -E "^\[sdt-pregraph\]|real"
( time soapdenovo-trans_amd/csrc/sdt-pregraph pregraph -s /tmp/r30.cfg -K 31 -p 16 -o /tmp/o3 --gpus 2 --share-device ) 2>&1 | grep -E "^\[sdt-pregraph\]|real"
cmp /tmp/o2.kmerFreq /tmp/o3.kmerFreq && cmp /tmp/o2.vertex /tmp/o3.vertex && cmp /tmp/o2.preArc /tmp/o3.preArc && echo "2-rank files identical"
