#!/bin/bash
# HBM bytes per launch of the bf16 step's kernels (FETCH_SIZE / WRITE_SIZE, one pass each) -> gpurun_out/pmcb/traffic.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcb
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --single-stream --no-kernel-profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --single-stream --no-kernel-profile > /dev/null 2>&1
python3 $ROOT/profiles/collect_traffic.py $OUT/fetch $OUT/write $OUT/hbm_traffic_bf16.json > $OUT/traffic.txt
rm -rf $OUT/fetch $OUT/write
