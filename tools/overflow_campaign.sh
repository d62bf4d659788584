#!/bin/bash
# Round-5 gradient overflow, repeated whole-iteration runs: N runs of 12 CRPS iterations with a one-rank RCCL
# group per configuration; a run "overflowed" when any parameter's second moment is non-finite at its end.
# usage: overflow_campaign.sh <outfile> <runs> <name>=<env assignments, comma separated> ...
#   e.g. overflow_campaign.sh out.txt 8 "memset=SWIFTK_TUNE=25:1,SWIFTK_MNB_MODE=clear" "shipped="
out=$1; runs=$2; shift 2
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  name=${cfg%%=*}; envs=${cfg#*=}; TB_ARGS=""; case "$envs" in *NOGROUP*) TB_ARGS="--dist 0"; envs=$(echo "$envs" | sed "s/,*NOGROUP//");; esac
  bad=0; fail=0
  for i in $(seq 1 "$runs"); do
    log=$(env $(echo "$envs" | tr ',' ' ') timeout 240 python tools/train_bench.py --loss crps --iters 12 $TB_ARGS 2>&1)
    rc=$?
    line=$(echo "$log" | grep "OVERFLOW-CHECK" | head -1)
    n=$(echo "$line" | sed -n 's/.*exp_avg_sq: \([0-9]*\):.*/\1/p')
    if [ -z "$n" ]; then fail=$((fail+1)); echo "[$name run $i] rc=$rc NO RESULT: $(echo "$log" | tail -2 | tr '\n' ' ')" >> "$out";
    elif [ "$n" != "0" ]; then bad=$((bad+1)); echo "[$name run $i] OVERFLOW $line" | cut -c1-300 >> "$out"; echo "$log" | grep "OVERFLOW-WHERE" | head -6 | cut -c1-400 >> "$out";
    else echo "[$name run $i] clean; $(echo "$log" | grep -o 'CRPS finetune.*s/iteration' | head -1)" >> "$out"; fi
  done
  echo "== $name ($envs): $bad of $runs runs overflowed, $fail without a result" >> "$out"
done
