set -e
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t16.log 2>&1 || (tail -40 gpurun_out/t16.log; exit 1)
tail -2 gpurun_out/t16.log
timeout -k 10 300 python tools/bench_configs.py c5 2>/dev/null | cut -c1-250
