#!/bin/bash
# Copies the outputs of collect_round.sh (gpurun_out/round/) into profiles/<tag>_*: bash profiles/copy_round.sh r03
set -e
TAG=${1:?tag}
R=gpurun_out/round
newest() { ls -t $1 2>/dev/null | head -1; }
cp $R/bench.json profiles/${TAG}_bench.json
grep -v amdgpu.ids $R/kernel_table.txt > profiles/${TAG}_kernel_table.txt
cp "$(newest "$R/prof_single/*/*_kernel_stats.csv")" profiles/${TAG}_kernel_stats_single_stream.csv
cp "$(newest "$R/prof_3streams/*/*_kernel_stats.csv")" profiles/${TAG}_kernel_stats_3streams.csv
cp $R/bench_single_stream_under_rocprof.json profiles/${TAG}_bench_single_stream_under_rocprof.json
cp $R/traffic.txt profiles/${TAG}_hbm_traffic.txt
cp $R/mfma_busy.txt profiles/${TAG}_mfma_busy.txt
cp $R/hbm_traffic.json $R/mfma_busy.json profiles/
cp $R/bench_bf16.json profiles/${TAG}_bf16_bench.json
grep -v amdgpu.ids $R/kernel_table_bf16.txt > profiles/${TAG}_bf16_kernel_table.txt
cp "$(newest "$R/prof_bf16/*/*_kernel_stats.csv")" profiles/${TAG}_bf16_kernel_stats_single_stream.csv
cp $R/configs.json profiles/${TAG}_config3_config4.json
ls -la profiles/${TAG}_*
