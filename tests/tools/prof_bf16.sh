#!/bin/bash
# One single-stream profile of bench.py (5 steps) -> gpurun_out/pb<tag>/ (run through gpurun).  usage: prof_bf16.sh [dtype] [tag]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pb$2
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pb$2 -- python3 $ROOT/bench.py --dtype ${1:-bf16} --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --single-stream > $ROOT/gpurun_out/pb$2.json 2> /dev/null
