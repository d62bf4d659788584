#!/bin/bash
# End-of-change profile set on the GPU box (run via gpurun from the repo root): tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the bench command (short run)      -> gpurun_out/<tag>_kernel_stats.csv
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE, separate, with --kernel-trace) -> gpurun_out/<tag>_pmc_summary.txt
# (rocprofv3 gets the program itself after `--`: python3 bench.py; TMPDIR=/tmp as the pool's recipe asks)
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -- python3 bench.py --steps 4 --warmup 1 --no-extras > /tmp/prof_$tag/stats.log 2>&1
f=$(find /tmp/prof_$tag/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
python tools/kstats.py $out/${tag}_kernel_stats.csv 16 | tee $out/${tag}_kernel_stats_top.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_$tag/fetch -- python3 bench.py --steps 2 --warmup 1 --no-extras > /tmp/prof_$tag/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_$tag/write -- python3 bench.py --steps 2 --warmup 1 --no-extras > /tmp/prof_$tag/write.log 2>&1
python tools/pmc_summary.py $(find /tmp/prof_$tag/fetch /tmp/prof_$tag/write -name "*counter_collection.csv") | tee $out/${tag}_pmc_summary.txt
