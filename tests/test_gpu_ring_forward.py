"""k_dense_fwd_ring (csrc/pyz_gemm_ring.h): the Dense forward of mid-size launches -- 32-row blocks, both operands
through an LDS-DMA ring -- against the float64 oracle, at the shape it is built for: 8 and 16 particles of
784 -> 200 -> 10 at batch 1024 (one rank's share of the sharded SVGD step), with and without the row gather,
ragged batches, odd particles (their [W; b] blocks are only 8-byte aligned: D = 159 010) and the batch copy the
workgroups leave for the weight-gradient kernel.  Every test asserts that the ring kernel is the one that ran."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mlp as o_mlp

from bayesian_inference_for_nn_amd import synth

MNIST = o_mlp.MLPSpec((784, 200, 10), ("relu", "softmax"), "scce")


def close(gpu, ref, rel=1e-4, what=""):
    gpu = np.asarray(gpu.detach().cpu().numpy() if hasattr(gpu, "detach") else gpu, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert gpu.shape == ref.shape, (what, gpu.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(gpu - ref).max()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e})"


@pytest.fixture(scope="module")
def eng(gpu_device):
    from bayesian_inference_for_nn_amd import engine
    return engine


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def _particles(P, seed):
    rng = np.random.default_rng(seed)
    base = synth.glorot_uniform(MNIST.dims)
    return (base[None, :] + 0.05 * rng.normal(size=(P, MNIST.n_params))).astype(np.float32)


def _ran(kp, name):
    return any(n.startswith(name) for n, _ in kp.launches)


@pytest.mark.parametrize("P,batch,gather", [(8, 1024, False), (8, 1000, True), (16, 1024, True), (7, 897, False)])
def test_ring_forward_matches_oracle(eng, P, batch, gather):
    """pyz_mlp_forward: softmax outputs of every particle, rows past a ragged batch untouched."""
    x, _ = synth.mnist_like(2048)
    rng = np.random.default_rng(5 + P)
    parts = _particles(P, 11 + P)
    plan = eng.MLPPlan(eng.MLPSpec(MNIST.dims, MNIST.acts, MNIST.loss), max_batch=1024, max_particles=P)
    idx = rng.permutation(2048)[:batch].astype(np.int32) if gather else None
    with eng.KernelProbe(16) as kp:
        out = plan.forward(dev(parts), dev(x), batch=batch, row_idx=dev(idx, torch.int32) if gather else None)
    assert _ran(kp, "k_dense_fwd_ring"), kp.launches
    xs = x[idx] if gather else x[:batch]
    for p in range(P):
        ref = o_mlp.predict(parts[p], xs, MNIST)
        close(out[p], ref, what=f"particle {p}")
        assert (out[p].argmax(dim=1).cpu().numpy() == ref.argmax(axis=1)).mean() > 0.999
    plan.close()


@pytest.mark.parametrize("P,batch", [(8, 1024), (8, 896)])
def test_ring_forward_feeds_the_gradient_pass(eng, P, batch):
    """pyz_mlp_loss_grad with gathered rows: the ring forward leaves the contiguous batch copy that k_wgrad_all reads
    (every workgroup stores its share of the slabs); losses and gradients of every particle against the oracle."""
    x, y = synth.mnist_like(2048)
    rng = np.random.default_rng(31)
    idx = rng.permutation(2048)[:batch].astype(np.int32)
    parts = _particles(P, 41)
    plan = eng.MLPPlan(eng.MLPSpec(MNIST.dims, MNIST.acts, MNIST.loss), max_batch=1024, max_particles=P)
    with eng.KernelProbe(16) as kp:
        loss, grad = plan.loss_grad(dev(parts), dev(x), dev(y, torch.int32), batch=batch, row_idx=dev(idx, torch.int32))
    assert _ran(kp, "k_dense_fwd_ring"), kp.launches
    for p in range(P):
        rl, rg, _ = o_mlp.loss_and_grad(parts[p], x[idx], y[idx], MNIST)
        close(loss[p:p + 1], [rl], what=f"loss {p}")
        close(grad[p], rg, what=f"gradient {p}")
    plan.close()
