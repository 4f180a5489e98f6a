#!/usr/bin/env python3
"""GPU-box check of the sharded SVGD path with real kernels:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/svgd_two_rank_check.py
Two ranks (gloo; both may sit on the single GPU of a test box) each own half of the particles,
all-gather the matrix every step (Jacobi sweep) and must reproduce the unsharded Jacobi run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bayesian_inference_for_nn_amd import parallel, synth  # noqa: E402
from bayesian_inference_for_nn_amd.datasets import Dataset  # noqa: E402
from bayesian_inference_for_nn_amd.distributions import GaussianPrior  # noqa: E402
from bayesian_inference_for_nn_amd.losses import SparseCategoricalCrossentropy  # noqa: E402
from bayesian_inference_for_nn_amd.nn import sequential_json  # noqa: E402
from bayesian_inference_for_nn_amd.optimizers import SVGD  # noqa: E402
from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters  # noqa: E402


def run(shard, steps=6):
    x, y = synth.moons(500, seed=42)
    ds = Dataset((x, y), SparseCategoricalCrossentropy, "Classification", seed=3)
    opt = SVGD()
    opt.compile(HyperParameters(lr=0.05, M=8, batch_size=100), sequential_json(2, [16, 2], ["relu", "softmax"]), ds,
                verbose=False, prior=GaussianPrior(0.0, 0.3), seed=2, shard=shard, sweep="jacobi")
    losses = [float(opt.step()) for _ in range(steps)]
    ens, _, _ = opt.result()
    return np.stack([np.concatenate([w.reshape(-1) for w in m.get_weights()]) for m in ens]), losses


def main():
    ngpu = torch.cuda.device_count()
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(rank % max(ngpu, 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    sharded, l_sh = run(True)
    assert parallel.world_info()[1] == 2
    whole, l_wh = run(False)
    err = float(np.abs(sharded - whole).max() / np.abs(whole).max())
    lerr = max(abs(a - b) / abs(b) for a, b in zip(l_sh, l_wh))
    print(f"rank {rank}: sharded (2 ranks x 4 particles, all-gather per step) vs whole-matrix Jacobi: "
          f"max rel particle diff {err:.2e}, max rel loss diff {lerr:.2e}", flush=True)
    ok = err < 1e-5 and lerr < 1e-5
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
