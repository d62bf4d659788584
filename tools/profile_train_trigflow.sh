#!/bin/bash
# Kernel statistics of a TrigFlow pre-training iteration: tools/profile_train_trigflow.sh <tag>
#   -> gpurun_out/<tag>_train_trigflow_top.txt (python3 itself after `--`, TMPDIR=/tmp)
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out
d=/tmp/proft_${tag}_tf; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/train_bench.py --loss trigflow --iters 6 --dist 0 > $d/run.log 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1)
{ tail -1 $d/run.log; python tools/kstats.py "$f" 30; } | tee $out/${tag}_train_trigflow_top.txt
