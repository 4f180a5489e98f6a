set -e
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t10.log 2>&1 || (tail -40 gpurun_out/t10.log; exit 1)
tail -2 gpurun_out/t10.log
L=bayesian_inference_for_nn_amd/csrc/libpyz.so
rm -f gpurun_out/variants13.jsonl
timeout -k 10 120 python tools/variants.py $L hook >> gpurun_out/variants13.jsonl 2>gpurun_out/variants13.err
cat gpurun_out/variants13.jsonl
timeout -k 10 200 python tools/stamps.py > gpurun_out/stamps16.txt 2>&1
