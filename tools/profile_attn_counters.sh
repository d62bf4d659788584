#!/bin/bash
# SQ counters of the fused to_qkv + window attention kernel: tools/profile_attn_counters.sh <tag>
# Three programs (fused, fused with the attention core skipped, the plain to_qkv GEMM) x two --pmc passes of <= 8 SQ counters
# each, --kernel-trace only beside --pmc (the pool refuses other trace domains with counters); python3 itself after `--`.
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out
root=$PWD
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $out/${tag}_sq_counter_names.txt
for mode in fused fused_nocore gemm; do
  for pass in 1 2; do
    d=/tmp/pac_${tag}_${mode}_$pass; rm -rf $d; mkdir -p $d
    if [ $pass = 1 ]; then P="$P1"; else P="$P2"; fi
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python3 $root/tools/attn_counters.py $mode 96 6 > $d/run.log 2>&1
    tail -2 $d/run.log
  done
done
cd $root
for mode in fused fused_nocore gemm; do
  echo "== $mode"
  python tools/pmc_summary.py $(find /tmp/pac_${tag}_${mode}_1 /tmp/pac_${tag}_${mode}_2 -name "*counter_collection.csv")
done | tee $out/${tag}_attn_counters.txt
python tools/attn_counters_report.py /tmp/pac_${tag}_fused /tmp/pac_${tag}_fused_nocore /tmp/pac_${tag}_gemm | tee $out/${tag}_attn_counters_report.txt
