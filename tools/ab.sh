#!/bin/bash
# A/B two builds of libswiftk on one box: tools/ab.sh <libA> <libB> [rounds]  (paths relative to the repo root)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    SWIFTK_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --cpu-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d.get('attention_roofline',{}).get('avg_launch_ms',0),4))"
  done
done
