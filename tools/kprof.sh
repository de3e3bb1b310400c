#!/bin/bash
# per-kernel times of a short bench run: tools/kprof.sh <tag> [bench args]   (run on the GPU box through gpurun)
tag=$1; shift
mkdir -p /root/repo/gpurun_out/r2/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2/$tag -- python /root/repo/bench.py --no-secondary --no-cpu-baseline --profile-run --steps 5 --warmup 2 "$@" > /root/repo/gpurun_out/r2/$tag/bench.json 2>/dev/null
cd /root/repo
db=$(ls gpurun_out/r2/$tag/*/*_results.db | head -1)
python tools/prof_summary.py $db 7 | head -16
