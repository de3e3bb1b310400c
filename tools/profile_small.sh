#!/bin/bash
# kernel trace + SQ counters of single AES-128 / SHA-256 proofs (LDS runs) -> gpurun_out/<tag>/  (run through gpurun)
tag=${1:-r03_small}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/tools/lat_small.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -- $CMD > $out/lat.txt 2>/dev/null
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/sq -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES -d $out/sq2 -- $CMD > /dev/null 2>&1
# HBM traffic (BASELINE config 3: rocprof HBM GB/s against peak): separate FETCH_SIZE / WRITE_SIZE passes, as for the headline
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/write -- $CMD > /dev/null 2>&1
cd /root/repo
# lat_small makes 5 + 40 + 20 = 65 proofs of each circuit
python tools/prof_summary.py $(ls $out/trace/*/*_results.db | head -1) 65 > $out/${tag}_kernel_stats.txt
python tools/prof_summary.py $(ls $out/sq/*/*_results.db | head -1) 65 | sed -n '/counters_collection/,$p' | grep -v "^# counters" | grep "k_interp_lds\|k_interp_narrow" > $out/${tag}_sq_counters.txt
python tools/prof_summary.py $(ls $out/sq2/*/*_results.db | head -1) 65 | sed -n '/counters_collection/,$p' | grep -v "^# counters" | grep "k_interp_lds\|k_interp_narrow" >> $out/${tag}_sq_counters.txt
python tools/pmc_summary.py $(ls $out/fetch/*/*_results.db | head -1) $(ls $out/write/*/*_results.db | head -1) 65 $out/${tag}_pmc_traffic > /dev/null
rm -rf $out/trace $out/sq $out/sq2 $out/fetch $out/write
grep LDS_RUN $out/lat.txt; head -8 $out/${tag}_kernel_stats.txt; cat $out/${tag}_sq_counters.txt
