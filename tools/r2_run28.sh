#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python tools/gen_fastq.py /tmp/r30 30000000 150 20000 > /dev/null 2>&1
cat /tmp/r30/reads.fq > /dev/null
for i in 1 2 3; do
( time SDT_TIMING=1 soapdenovo-trans_amd/csrc/sdt-pregraph pregraph -s /tmp/r30/lib.cfg -K 31 -p 16 -o /tmp/o1 --hash-only ) 2>&1 | grep -E "sdt-pregraph\]|real|libsdt"
done
