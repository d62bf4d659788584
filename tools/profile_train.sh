#!/bin/bash
# Kernel statistics of the two training iterations on the GPU box: tools/profile_train.sh <tag>
#   rocprofv3 --kernel-trace --stats of tools/train_bench.py (CRPS finetune, BASELINE config 5; sCM, config 4)
#   -> gpurun_out/<tag>_train_{crps,scm}_kernel_stats.csv + _top.txt (python3 itself after `--`, TMPDIR=/tmp)
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out
for loss in crps scm; do
  d=/tmp/proft_${tag}_$loss; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/train_bench.py --loss $loss --iters 4 > $d/run.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${tag}_train_${loss}_kernel_stats.csv
  { tail -2 $d/run.log; python tools/kstats.py $out/${tag}_train_${loss}_kernel_stats.csv 24; } | tee $out/${tag}_train_${loss}_top.txt
done
