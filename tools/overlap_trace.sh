#!/bin/bash
# kernel timeline of one proof with the mask generator beside the level launches: tools/overlap_trace.sh <tag> [ENV=V ...]
tag=$1; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp; cd /root/repo
env "$@" rocprofv3 --kernel-trace -d $out -- python /root/repo/tools/one_proof.py > $out/run.log 2>&1

db=$(ls $out/*/*_results.db | head -1)
python tools/trace_proof.py $db > $out/timeline.txt
head -80 $out/timeline.txt
