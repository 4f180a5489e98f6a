set -e
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1 || (tail -40 gpurun_out/t6.log; exit 1)
tail -2 gpurun_out/t6.log
L=bayesian_inference_for_nn_amd/csrc/libpyz.so
rm -f gpurun_out/variants8.jsonl
timeout -k 10 120 python tools/variants.py $L base >> gpurun_out/variants8.jsonl 2>gpurun_out/variants8.err
timeout -k 10 120 python tools/variants.py bayesian_inference_for_nn_amd/csrc/variants/libpyz_nt.so nt >> gpurun_out/variants8.jsonl 2>gpurun_out/variants8.err
cat gpurun_out/variants8.jsonl
timeout -k 10 200 python tools/stamps.py > gpurun_out/stamps12.txt 2>&1
