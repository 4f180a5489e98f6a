#!/bin/bash
# GPU box: the measurements that back profiles/ (run from the repo root through gpurun).
#   1. python bench.py                                  -> bench.json            (default invocation)
#      python bench.py --gpus 1 --steps 20 --warmup 5   -> bench_driver.json     (the driver's invocation)
#   2. rocprofv3 --kernel-trace --stats of both         -> kernel_stats*.csv + bench*_under_rocprof.json
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes    -> pmc_fetch/, pmc_write/  (tools/pmc_traffic.py reads them)
#   4. tools/bench_configs.py                            -> configs.jsonl
#   5. tools/bench_api.py                                -> bench_api.jsonl
#   6. bench.py --method svgd (both sweeps)              -> svgd_*.json
#   usage: tools/run_profiles.sh <tag> [a|b|all]   (a = 1 - 3, b = 4 - 6: each half fits one gpurun call)
set -e
OUT=gpurun_out/prof_$1
PART=${2:-all}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
if [ "$PART" != "b" ]; then
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2>$OUT/bench.err
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>$OUT/bench_driver.err
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2>$R/$OUT/rocprof_stats.err)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_driver -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$OUT/bench_driver_under_rocprof.json 2>$R/$OUT/rocprof_stats_driver.err)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-roofline > $R/$OUT/pmc_fetch.json 2>$R/$OUT/pmc_fetch.err)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-roofline > $R/$OUT/pmc_write.json 2>$R/$OUT/pmc_write.err)
fi
if [ "$PART" != "a" ]; then
timeout -k 10 600 python3 tools/bench_configs.py c1 c3 c4 c5 c5shard > $OUT/configs.jsonl 2>$OUT/configs.err
timeout -k 10 300 python3 tools/bench_api.py > $OUT/bench_api.jsonl 2>$OUT/bench_api.err
timeout -k 10 200 python3 bench.py --method svgd --steps 100 --warmup 10 > $OUT/svgd_gauss_seidel.json 2>$OUT/svgd_gs.err
timeout -k 10 200 python3 bench.py --method svgd --sweep jacobi --steps 100 --warmup 10 > $OUT/svgd_jacobi.json 2>$OUT/svgd_j.err
fi
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -60
cat $OUT/bench.json 2>/dev/null || true
cat $OUT/bench_driver.json 2>/dev/null || true
