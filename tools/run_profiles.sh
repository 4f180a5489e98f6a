#!/bin/bash
# GPU box: the measurements that back profiles/ (run from the repo root through gpurun).
#   1. python bench.py                                  -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same     -> kernel_stats.csv + bench_under_rocprof.json
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes    -> pmc_fetch/, pmc_write/  (tools/pmc_traffic.py reads them)
#   4. tools/bench_configs.py                            -> configs.jsonl
#   5. tools/bench_api.py                                -> bench_api.jsonl
set -e
OUT=gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2>$OUT/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>$OUT/rocprof_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.json 2>$OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-roofline > $OUT/pmc_write.json 2>$OUT/pmc_write.err
timeout -k 10 600 python3 tools/bench_configs.py > $OUT/configs.jsonl 2>$OUT/configs.err
timeout -k 10 300 python3 tools/bench_api.py > $OUT/bench_api.jsonl 2>$OUT/bench_api.err
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -40
cat $OUT/bench.json
