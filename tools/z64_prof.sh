#!/bin/bash
# kernel trace + SQ counters of the Z64 level kernels (k_z64_c4 / k_z64_fused): tools/z64_prof.sh <tag> [n_mul]  (run through gpurun)
tag=${1:-z64prof}
n=${2:-250000}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export RV_Z64_C4=$v
  CMD="python /root/repo/tools/z64_phases.py $n"
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace$v -- $CMD > $out/phases$v.txt 2>/dev/null
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/sq$v -- $CMD > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $out/sq2$v -- $CMD > /dev/null 2>&1
  ( cd /root/repo
    python tools/prof_summary.py $(ls $out/trace$v/*/*_results.db | head -1) 5 | head -8
    python tools/prof_summary.py $(ls $out/sq$v/*/*_results.db | head -1) 5 | sed -n '/counters_collection/,$p' | grep "k_z64"
    python tools/prof_summary.py $(ls $out/sq2$v/*/*_results.db | head -1) 5 | sed -n '/counters_collection/,$p' | grep "k_z64" ) > $out/${tag}_c4_$v.txt
  rm -rf $out/trace$v $out/sq$v $out/sq2$v
  cat $out/phases$v.txt | tail -1; cat $out/${tag}_c4_$v.txt
done
