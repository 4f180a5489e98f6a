set -e
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t12.log 2>&1 || (tail -40 gpurun_out/t12.log; exit 1)
tail -2 gpurun_out/t12.log
bash tools/run_profiles.sh v6
