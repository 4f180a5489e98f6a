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


C4 = o_mlp.MLPSpec((784, 400, 400, 10), ("relu", "relu", "softmax"), "scce")


@pytest.mark.parametrize("P,rows,gather", [(1, 6000, False), (1, 5989, True), (3, 2048, True)])
def test_ring_forward_on_column_groups(eng, P, rows, gather, monkeypatch):
    """Layers wider than the 200 columns a workgroup takes (BASELINE configs[3], 784 -> 400 -> 400 -> 10: the 6 000-row
    validation forward of BBB.py:203-209) are cut into column groups of 200; both hidden layers through the ring, ragged
    row counts, gathered rows, several particles -- against the oracle (to
    float32 summation order)."""
    monkeypatch.setenv("PYZ_FWD_RING_WIDE", "1")     # (opt-in: no faster than k_dense_fwd at this shape, pyz_gemm_ring.h)
    n = 6000
    x, y = synth.mnist_like(n)
    rng = np.random.default_rng(7 + rows)
    base = synth.glorot_uniform(C4.dims)
    parts = (base[None, :] + 0.02 * rng.normal(size=(P, C4.n_params))).astype(np.float32)
    plan = eng.MLPPlan(eng.MLPSpec(C4.dims, C4.acts, C4.loss), max_batch=n, max_particles=P)
    idx = rng.permutation(n)[:rows].astype(np.int32) if gather else None
    with eng.KernelProbe(16) as kp:
        out = plan.forward(dev(parts), dev(x), batch=rows, row_idx=dev(idx, torch.int32) if gather else None)
    ring = [nm for nm, _ in kp.launches if nm.startswith("k_dense_fwd_ring")]
    assert len(ring) == 2, kp.launches
    xs = x[idx] if gather else x[:rows]
    for p in range(P):
        ref = o_mlp.predict(parts[p], xs, C4)
        close(out[p], ref, what=f"particle {p}")
    if P > 1:      # and as the first kernels of a gradient pass (the batch copy is left by the first column group only)
        loss, grad = plan.loss_grad(dev(parts), dev(x), dev(y, torch.int32), batch=rows, row_idx=dev(idx, torch.int32))
        for p in range(P):
            rl, rg, _ = o_mlp.loss_and_grad(parts[p], x[idx], y[idx], C4)
            close(loss[p:p + 1], [rl], what=f"loss {p}")
            close(grad[p], rg, rel=2e-4, what=f"gradient {p}")
    plan.close()
