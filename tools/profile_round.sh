#!/bin/bash
# Profiles of the benchmarked configuration at HEAD -> gpurun_out/<tag>/ (copy the summaries into profiles/):
#   tools/profile_round.sh r02        (run on the GPU box: gpurun -- tools/profile_round.sh r02)
tag=$1
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
W=2; K=6; N=$((W+K))
B="python /root/repo/bench.py --no-secondary --no-cpu-baseline --profile-run --steps $K --warmup $W"
# kernel trace twice: with RV_EARLY=0 (every kernel at its own duration) and as shipped.  Under rocprofv3 the runtime turns
# copy-engine transfers into shader blits (__amd_rocclr_copyBuffer), and the early-corrections path of rv_prove keeps 160 MB of
# them in flight per proof: as blits they are not dispatched before the hash kernels and then share the chip with them, so the
# "early" trace shows the path's kernels (k_pack_corr_all, k_publish, k_copy_gaps) but overstates the hash kernels
RV_EARLY=0 rocprofv3 --kernel-trace --stats -d $out/trace -- $B > $out/bench_trace.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $out/trace_early -- $B > $out/bench_trace_early.json 2>/dev/null
# round 5: the mask generator runs BESIDE the level launches (RV_OVERLAP); the third trace is the round-4 schedule, every kernel alone
RV_EARLY=0 RV_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $out/trace_serial -- $B > $out/bench_trace_serial.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -- $B > /dev/null 2>&1
RV_EARLY=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/sq -- $B > /dev/null 2>&1
cd /root/repo
python tools/prof_summary.py $(ls $out/trace/*/*_results.db | head -1) $N > $out/${tag}_bench_kernel_stats.txt
sed -i '1a # RV_EARLY=0, RV_OVERLAP as shipped: k_aes_gf2_masks_col4 runs on a stream of its own beside the k_interp_full launches, so both kinds of kernel show their CO-RESIDENT durations (the _serial file has them alone).  (rv_prove without the early-corrections path: its copy-engine transfers become shader blits under rocprofv3 and distort the hash kernels -- see the _early file)' $out/${tag}_bench_kernel_stats.txt
python tools/prof_summary.py $(ls $out/trace_early/*/*_results.db | head -1) $N > $out/${tag}_bench_kernel_stats_early.txt
sed -i '1a # as shipped (early corrections on).  __amd_rocclr_copyBuffer = the path\x27s 160 MB of copy-engine transfers per proof as the shader blits rocprofv3 turns them into: they run beside the hash kernels here (k_b3_* times are NOT what a proof sees without the profiler); k_pack_corr_all / k_publish / k_copy_gaps are the path\x27s own kernels' $out/${tag}_bench_kernel_stats_early.txt
python tools/prof_summary.py $(ls $out/trace_serial/*/*_results.db | head -1) $N > $out/${tag}_bench_kernel_stats_serial.txt
sed -i '1a # RV_EARLY=0 RV_OVERLAP=0: the mask generator before the first level, nothing beside the level launches (every kernel at its stand-alone duration)' $out/${tag}_bench_kernel_stats_serial.txt
python tools/pmc_summary.py $(ls $out/fetch/*/*_results.db | head -1) $(ls $out/write/*/*_results.db | head -1) $N $out/${tag}_pmc_traffic > /dev/null
python tools/prof_summary.py $(ls $out/sq/*/*_results.db | head -1) $N | sed -n '/counters_collection/,$p' > $out/${tag}_sq_counters.txt
python tools/sq_summary.py $(ls $out/sq/*/*_results.db | head -1) $N $out/${tag}_sq_counters.json > /dev/null
rm -rf $out/trace $out/trace_early $out/trace_serial $out/fetch $out/write $out/sq
head -12 $out/${tag}_bench_kernel_stats.txt; head -12 $out/${tag}_pmc_traffic.txt
