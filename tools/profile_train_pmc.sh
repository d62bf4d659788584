#!/bin/bash
# HBM traffic of the training-side streaming kernels: two --pmc passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) over
# the micro-benchmarks of the one-kernel ModulatedNorm backward and the pair-form ModulatedNorm tangent: tools/profile_train_pmc.sh <tag>
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out
for t in modnorm_bwd_bench modnorm_jvp_bench; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmc_${tag}_${t}_$c; rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/$t.py 8 > $d.log 2>&1
  done
done
python tools/pmc_summary.py $(find /tmp/pmc_${tag}_* -name "*counter_collection.csv") | tee $out/${tag}_train_pmc_summary.txt
