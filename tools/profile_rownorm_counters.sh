#!/bin/bash
# SQ counters of the complete-row wo / w2 + norm kernel against the tiled GEMM at 96 units ("a measured
# negative with counters"): tools/profile_rownorm_counters.sh <tag>.  Two --pmc passes of <= 9 SQ counters per program,
# --kernel-trace only beside --pmc; python3 itself after `--`.
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out
root=$PWD
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
cd /tmp
dirs=""
for mode in rownorm gemm; do
  for which in wo w2; do
    for pass in 1 2; do
      d=/tmp/prc_${tag}_${mode}_${which}_$pass; rm -rf $d; mkdir -p $d
      if [ $pass = 1 ]; then P="$P1"; else P="$P2"; fi
      rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python3 $root/tools/rownorm_counters.py $mode $which 96 6 > $d/run.log 2>&1
      tail -1 $d/run.log
    done
    dirs="$dirs /tmp/prc_${tag}_${mode}_${which}"
  done
done
cd $root
python tools/attn_counters_report.py $dirs | tee $out/${tag}_rownorm_counters_report.txt
