#!/bin/bash
# Collects the round's judged measurements on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect_round.sh
# Outputs under gpurun_out/round/, copied into profiles/ afterwards (see profiles/README.md).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_single $OUT/prof_3streams $OUT/prof_bf16
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --single-stream > $OUT/bench_single_stream_under_rocprof.json 2> /dev/null
echo "rocprof single done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_3streams -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 > $OUT/bench_3streams_under_rocprof.json 2> /dev/null
echo "rocprof 3 streams done"
# HBM traffic counters over the SAME launches the roofline averages over (all launches of each kernel in bench's
# train steps, one stream), one counter per pass as MI355X_MICROARCH.md prescribes
rm -rf $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --single-stream --no-kernel-profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --single-stream --no-kernel-profile > /dev/null 2>&1
python3 $ROOT/profiles/collect_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/hbm_traffic.json > $OUT/traffic.txt
# matrix-pipe busy cycles per kernel (north_star: "rocprof HBM GB/s and MFMA-busy"): SQ + GRBM counters, own pass
rm -rf $OUT/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --single-stream --no-kernel-profile > /dev/null 2>&1
python3 $ROOT/profiles/collect_mfma.py $OUT/pmc_mfma $OUT/mfma_busy.json > $OUT/mfma_busy.txt
echo "pmc done"
# the counter tables go where bench.py reads them (stamped with the digest of the sources they were collected on), THEN the
# fp32 headline (BASELINE configs[1]) incl. CPU baseline and the sustained window: its line carries traffic and mfma_busy
cp $OUT/hbm_traffic.json $OUT/mfma_busy.json $ROOT/profiles/
python3 $ROOT/bench.py --steps 100 --warmup 20 --kernel-table > $OUT/bench.json 2> $OUT/kernel_table.txt
echo "bench fp32 done"
# bf16 mixed precision (BASELINE configs[4], single-GPU share): bench line + kernel table + rocprof stats
python3 $ROOT/bench.py --dtype bf16 --steps 100 --warmup 20 --kernel-table --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/kernel_table_bf16.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16 -- python3 $ROOT/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --single-stream > $OUT/bench_bf16_single_stream_under_rocprof.json 2> /dev/null
echo "bf16 done"
# the other single-GPU configurations (config 3's per-GPU workload, config 4)
python3 $ROOT/profiles/time_configs.py > $OUT/configs.json 2> /dev/null
ls $OUT
