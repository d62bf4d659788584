#!/bin/bash
# SQ counters of the weight-gradient GEMM (gemm_tn_kernel), ping-pong k-loop off / on (tuning key 22): tools/profile_tn_counters.sh <tag> [shape]
# Two --pmc passes of <= 8 SQ counters each per arm, --kernel-trace only beside --pmc; python3 itself after `--`.
tag=$1; shape=${2:-w1}
export TMPDIR=/tmp
out=$PWD/gpurun_out
root=$PWD
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
cd /tmp
for pp in 0 1; do
  for pass in 1 2; do
    d=/tmp/ptc_${tag}_${shape}_pp${pp}_$pass; rm -rf $d; mkdir -p $d
    if [ $pass = 1 ]; then P="$P1"; else P="$P2"; fi
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python3 $root/tools/tn_counters.py $shape $pp 8 6 > $d/run.log 2>&1
    tail -1 $d/run.log
  done
done
cd $root
python tools/attn_counters_report.py /tmp/ptc_${tag}_${shape}_pp0 /tmp/ptc_${tag}_${shape}_pp1 | tee $out/${tag}_tn_${shape}_sq_counters_report.txt
