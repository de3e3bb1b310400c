#!/bin/bash
# Profiles of an arbitrary command -> gpurun_out/<tag>/<tag>_{kernel_stats,pmc_traffic}.txt (copy into profiles/):
#   tools/profile_cmd.sh <tag> <proofs the command makes> <command ...>     (run on the GPU box through gpurun)
# Every rocprofv3 pass runs under its own timeout (a pass that hangs must not take the box's whole limit with it).
tag=$1; N=$2; shift 2
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
T=${PROFILE_TIMEOUT:-300}
timeout $T rocprofv3 --kernel-trace --stats -d $out/trace -- "$@" > $out/cmd_trace.txt 2>/dev/null
timeout $T rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- "$@" > /dev/null 2>&1
timeout $T rocprofv3 --pmc WRITE_SIZE -d $out/write -- "$@" > /dev/null 2>&1
cd /root/repo
python tools/prof_summary.py $(ls $out/trace/*/*_results.db | head -1) $N > $out/${tag}_kernel_stats.txt
python tools/pmc_summary.py $(ls $out/fetch/*/*_results.db | head -1) $(ls $out/write/*/*_results.db | head -1) $N $out/${tag}_pmc_traffic > /dev/null
rm -rf $out/trace $out/fetch $out/write
head -14 $out/${tag}_kernel_stats.txt; head -14 $out/${tag}_pmc_traffic.txt
