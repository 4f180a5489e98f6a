set -e
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t4.log 2>&1 || (tail -30 gpurun_out/t4.log; exit 1)
tail -2 gpurun_out/t4.log
timeout -k 10 200 python bench.py --steps 4096 --warmup 512 > gpurun_out/bench_v6.json 2>gpurun_out/bench_v6.err
cat gpurun_out/bench_v6.json
