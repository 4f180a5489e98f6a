"""End-to-end tests of the drop-in Python surface on the GPU: the reference's call pattern
(compile / train / step / result / predict) driven through the HIP kernels, checked against
the CPU oracle where a deterministic comparison exists."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import mlp as o_mlp
from oracle import philox as o_philox
from oracle import predict as o_predict
from oracle import sgld as o_sgld
from oracle import bbb as o_bbb

from bayesian_inference_for_nn_amd import synth
from bayesian_inference_for_nn_amd.datasets import Dataset
from bayesian_inference_for_nn_amd.distributions import GaussianPrior
from bayesian_inference_for_nn_amd.losses import MeanSquaredError, SparseCategoricalCrossentropy
from bayesian_inference_for_nn_amd.nn import BayesianModel, model_from_json, sequential_json
from bayesian_inference_for_nn_amd.optimizers import BBB, HMC, SGD, SGLD, SVGD, SWAG
from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters


@pytest.fixture(scope="module", autouse=True)
def _gpu(gpu_device):
    return gpu_device


def moons_dataset(n=500, seed=3):
    x, y = synth.moons(n, seed=42)
    return Dataset((x, y), SparseCategoricalCrossentropy, "Classification", seed=seed)


MOONS_JSON = sequential_json(2, [16, 2], ["relu", "softmax"])


def test_sgd_regression_example_converges():
    """simple_regression_example.py: y = 2x + 2, Dense(1), SGD(lr=1e-3, frequency=1)."""
    x, y = synth.linreg(600)
    ds = Dataset((x, y), MeanSquaredError, "Regression", seed=0)
    cfg = sequential_json(1, [1], ["linear"])
    start = model_from_json(cfg)
    start.set_weights([np.array([[0.3]]), np.array([-0.1])])
    opt = SGD()
    opt.compile(HyperParameters(lr=1e-3, frequency=1), cfg, ds, verbose=False, starting_model=start, seed=1)
    first = float(opt.step())
    opt.train(3000)
    bm = opt.result()
    assert isinstance(bm, BayesianModel)
    w, b = bm._model.layers[0].trainable_variables
    assert abs(w[0, 0] - 2.0) < 0.1 and abs(float(opt.step())) < first / 50
    xt, yt = next(iter(ds.test_data.batch(ds.test_size)))
    samples, mean = bm.predict(xt, nb_samples=3)
    assert len(samples) == 3 and mean.shape == (ds.test_size, 1)
    np.testing.assert_allclose(mean, xt.numpy() * w[0, 0] + b[0], rtol=1e-5, atol=1e-4)     # Deterministic posterior
    with pytest.raises(Exception, match="Model Already compiled"):
        opt.compile(HyperParameters(lr=1e-3, frequency=1), cfg, ds, starting_model=start)


def test_sgld_surface_matches_oracle_and_resident_training():
    ds = moons_dataset()
    hyp = HyperParameters(lr_upper=0.01, lr_lower=0.003, lr_gamma=0.99, batch_size=96)
    n_steps = 11                                      # 400 train rows / 96 -> ragged batch of 16 every 5th step
    opt = SGLD()
    opt.compile(hyp, MOONS_JSON, ds, verbose=False, seed=123)
    theta0 = opt._theta.cpu().numpy().copy()
    # oracle replay of the same batches and the same Philox noise stream
    spec = o_mlp.MLPSpec((2, 16, 2), ("relu", "softmax"), "scce")
    x, y = ds.train_data.as_numpy()
    st = o_sgld.SGLDState(theta0)
    lr = o_sgld.lr_schedule(n_steps, 0.01, 0.003, 0.99)
    opt._nb_iterations = n_steps
    opt._init_sgld_lr()
    rets = []
    for s in range(n_steps):
        ret = opt.step()
        pos, b = opt._pos, None
        perm = opt._perm_dev.cpu().numpy()
        rows = perm[pos - (96 if pos % 96 == 0 else pos % 96):pos]
        _, r = o_sgld.sgld_step(st, x[rows], y[rows], spec, lr(s), o_philox.normal(opt._seed, 0, s, spec.n_params))
        rets.append((float(ret), r))
    for got, ref in rets:
        assert abs(got - ref) <= 1e-4 * abs(ref)
    np.testing.assert_allclose(opt._theta.cpu().numpy(), st.theta, rtol=0, atol=1e-4 * np.abs(st.theta).max())
    bm = opt.result()
    np.testing.assert_allclose(bm._distributions[0]._tf_distribution.loc, st.mean[:48], atol=1e-5)
    np.testing.assert_allclose(bm._distributions[1]._tf_distribution.scale, (st.sq_mean - st.mean ** 2)[48:], atol=1e-6)
    # verbose=False training runs device-resident (hipGraph) and must agree with the stepwise path
    a, b2 = SGLD(), SGLD()
    a.compile(hyp, MOONS_JSON, ds, verbose=False, seed=7)
    b2.compile(hyp, MOONS_JSON, ds, verbose=False, seed=7)
    a._resident_chunks = (40, 0.5, 40)                          # planned and launched as 40 + 20 + 10 steps
    a.train(70)
    # train() is synchronous like the reference's loop: the HOST has joined the run stream when it returns (a wait of
    # the caller's stream enqueued behind the run instead slows every step of the run, Optimizer._join_run)
    assert a._res_stream.query(), "train() returned with its run still in flight"
    stream_of_first_call = a._res_stream
    b2._nb_iterations = 70
    b2._init_sgld_lr()
    for _ in range(70):
        b2.step()
    ta, tb = a._theta.cpu().numpy(), b2._theta.cpu().numpy()
    np.testing.assert_allclose(ta, tb, rtol=0, atol=2e-5 * np.abs(tb).max())
    assert a._n == 70 and abs(float(a._running_dev.item()) - float(b2._running_dev.item())) < 1e-3
    a.train(10)
    assert a._res_stream is stream_of_first_call and a._res_stream.query()   # one run stream per optimizer


def test_hmc_surface_bookkeeping_and_nan_prior():
    import random
    random.seed(0)
    ds = moons_dataset()
    opt = HMC()
    opt.compile(HyperParameters(epsilon=0.002, m=0.5, L=8), MOONS_JSON, ds, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=5)
    assert opt._batch_size == 400 and float(opt._q.abs().max()) == 0.0          # q <- prior mean
    opt.train(12)
    assert sum(opt._frequency) == 13 and len(opt._samples) == len(opt._frequency)   # starting q + 12 proposals
    assert opt._total_runs == 12 and 0 < opt._accepted_runs <= 12
    bm = opt.result()
    assert bm._layers_dtbn_intervals == [[0, 1]]
    xt, yt = next(iter(ds.test_data.batch(ds.test_size)))
    _, mean = bm.predict(xt, nb_samples=20)
    np.testing.assert_allclose(mean.sum(axis=1), 1.0, atol=1e-5)
    # negative rho (HMC_classification.py:50): NaN potential -> every non-burn proposal rejected
    opt2 = HMC()
    opt2.compile(HyperParameters(epsilon=0.005, m=0.5, L=3), MOONS_JSON, ds, verbose=False, prior=GaussianPrior(0.0, -1.0), seed=5)
    opt2.train(6)
    assert opt2._accepted_runs == 0 and opt2._frequency == [7] and len(opt2._samples) == 1
    # the kwarg-name quirk of HMC.py:61-62 is reproduced
    with pytest.raises(KeyError):
        HMC().compile(HyperParameters(epsilon=0.005, m=0.5, L=3), MOONS_JSON, ds, verbose=False,
                      prior=GaussianPrior(0.0, 1.0), nb_burn_epoch=3)
    # independent chains in one launch
    opt3 = HMC()
    opt3.compile(HyperParameters(epsilon=0.005, m=0.5, L=4), MOONS_JSON, ds, verbose=False, prior=GaussianPrior(0.0, 1.0),
                 seed=5, n_chains=3)
    opt3.train(4)
    assert all(sum(f) == 5 for f in opt3._chain_freq)
    assert not torch.equal(opt3._q[0], opt3._q[1])


def test_bbb_surface_matches_oracle_and_result_tuple():
    ds = moons_dataset()
    opt = BBB()
    opt.compile(HyperParameters(lr=0.05, alpha=0.1, batch_size=128, pi=0.75), MOONS_JSON, ds, verbose=False,
                prior=GaussianPrior(0.0, -1.0), prior2=GaussianPrior(0.5, 0.5), seed=9)
    pm, pr = o_bbb.mix_prior(0.0, -1.0, 0.5, 0.5, 0.75)
    assert abs(opt._prior._mean - pm) < 1e-12 and abs(opt._prior._std_dev - pr) < 1e-12
    spec = o_mlp.MLPSpec((2, 16, 2), ("relu", "softmax"), "scce")
    x, y = ds.train_data.as_numpy()
    mu, rho = opt._mu.cpu().numpy().astype(np.float64), opt._rho.cpu().numpy().astype(np.float64)
    assert np.all(mu == pm) and np.allclose(rho, pr)
    for s in range(1, 13):
        cost = opt.step()
        pos = opt._pos
        perm = opt._perm_dev.cpu().numpy()
        rows = perm[pos - (128 if pos % 128 == 0 else pos % 128):pos]
        out = o_bbb.bbb_step(mu, rho, o_philox.normal(opt._seed, 1, s, spec.n_params), x[rows], y[rows], spec, 0.05, 0.1, pm, pr)
        mu, rho = out["mu"], out["rho"]
        assert abs(float(cost) - out["cost"]) <= 1e-4 * abs(out["cost"]) + 1e-5
    np.testing.assert_allclose(opt._mu.cpu().numpy(), mu, atol=1e-4 * np.abs(mu).max())
    np.testing.assert_allclose(opt._rho.cpu().numpy(), rho, atol=1e-4 * np.abs(rho).max())
    assert len(opt.train_losses) == 11 and len(opt.val_losses) == 11          # every step except step 10 (BBB.py:203)
    vx, vy = ds.valid_data.as_numpy()
    ref_val = o_bbb.validation_loss(opt._w.cpu().numpy(), vx, vy, spec)
    assert abs(float(opt.val_losses[-1]) - ref_val) <= 1e-4 * ref_val
    res = opt.result()
    model, tl, vl = res
    assert isinstance(model, BayesianModel) and tl is opt.train_losses and vl is opt.val_losses
    xt, _ = next(iter(ds.test_data.batch(ds.test_size)))
    _, mean = res.predict(xt, nb_samples=8)             # attribute access forwards to the model (BBB_mnist.py:59)
    assert mean.shape == (ds.test_size, 2)
    np.testing.assert_allclose(model._distributions[1]._tf_distribution.scale, o_bbb.softplus(rho)[48:], rtol=1e-4)


def test_list_priors_through_the_surface():
    ds = moons_dataset()
    opt = BBB()
    opt.compile(HyperParameters(lr=0.05, alpha=0.1, batch_size=128), MOONS_JSON, ds, verbose=False,
                prior=GaussianPrior([0.0, 0.1], [-1.0, 0.5]), seed=4)
    assert float(opt._mu[:48].abs().max()) == 0.0 and abs(float(opt._mu[60]) - 0.1) < 1e-7     # posterior <- prior, per layer
    costs = [float(opt.step()) for _ in range(20)]
    assert np.isfinite(costs).all() and not torch.equal(opt._mu[:48], torch.zeros(48, device="cuda"))
    h = HMC()
    h.compile(HyperParameters(epsilon=0.002, m=0.5, L=4), MOONS_JSON, ds, verbose=False, prior=GaussianPrior([0.0, 0.0], [1.0, 2.0]), seed=4)
    h.train(5)
    assert sum(h._frequency) == 6


def test_svgd_surface():
    ds = moons_dataset()
    opt = SVGD()
    opt.compile(HyperParameters(lr=0.05, M=4, batch_size=100), MOONS_JSON, ds, verbose=False, prior=GaussianPrior(0.0, 1.0), seed=2)
    assert opt._all.shape == (4, 82) and opt._sweep == "gauss_seidel"
    p0 = opt._all.clone()
    assert abs(float(p0.std()) - 1.0) < 0.2
    losses = [float(opt.step()) for _ in range(20)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert len(opt.train_losses) == 2 and len(opt.valid_losses) == 2          # every 10 steps (SVGD.py:137-139)
    assert not torch.equal(p0, opt._all)
    ensemble, tl, vl = opt.result()
    assert len(ensemble) == 4
    xt, yt = next(iter(ds.test_data.batch(ds.test_size)))
    agg = sum(m.predict(xt) for m in ensemble) / len(ensemble)
    np.testing.assert_allclose(agg, ensemble.predict(xt), rtol=1e-6)
    np.testing.assert_allclose(ensemble[1].get_weights()[0].reshape(-1), opt._all[1, :32].cpu().numpy())


def test_bayesian_model_predict_matches_oracle():
    cfg = sequential_json((4, 4), [12, 3], ["tanh", "softmax"])
    bm = BayesianModel(cfg)
    from bayesian_inference_for_nn_amd.distributions import Sampled
    rng = np.random.default_rng(0)
    D = bm._model.count_params()
    cand = [(rng.normal(size=D) * 0.4).astype(np.float32) for _ in range(5)]
    bm.apply_distribution(Sampled(cand, [1, 2, 3, 1, 1]), 0, 2)
    x = rng.normal(size=(37, 4, 4)).astype(np.float32)
    W = bm.sample_weights_matrix(6)
    bm.sample_weights_matrix = lambda n: W[:n]
    samples, mean = bm.predict(x, nb_samples=6)
    spec = o_mlp.MLPSpec((16, 12, 3), ("tanh", "softmax"), "scce")
    rs, rm = o_predict.predict(W, x.reshape(37, -1), spec)
    np.testing.assert_allclose(np.stack(samples), rs, atol=1e-5)
    np.testing.assert_allclose(mean, rm, atol=1e-5)
    assert np.array_equal(np.argmax(mean, axis=1), np.argmax(rm, axis=1))          # integer class labels bit-exact
    # more rows than one launch sequence takes (and fewer samples per launch than asked for): the pieces are joined on
    # the device and must give the same arrays
    bm._predict_rows_cap, bm._plan = 16, None
    os_ws = os.environ.get("PYZ_PREDICT_WS")
    os.environ["PYZ_PREDICT_WS"] = str(16 * 2 * (12 + 3) * 4)                      # four samples per launch
    try:
        samples2, mean2 = bm.predict(x, nb_samples=6)
    finally:
        if os_ws is None:
            del os.environ["PYZ_PREDICT_WS"]
        else:
            os.environ["PYZ_PREDICT_WS"] = os_ws
    np.testing.assert_array_equal(np.stack(samples2), np.stack(samples))
    np.testing.assert_array_equal(np.asarray(mean2), np.asarray(mean))


def test_compat_standins_run_a_reference_style_driver():
    """compat/: `import tensorflow as tf` + `from Pyesian...` resolve to the stand-ins and a driver written in
    the reference's style runs end to end on the GPU (run in-process: no exec after the GPU is initialised)."""
    import os
    import runpy
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        runpy.run_path(os.path.join(root, "examples", "hmc_classification_compat.py"), run_name="__main__")
    finally:
        sys.path.remove(os.path.join(root, "compat"))


def test_swag_moons_posterior_and_store_load(tmp_path):
    """SWAG from an SGD-pretrained model (the reference's SWAG usage): per-layer
    MultivariateNormalDiagPlusLowRank posterior, store/load round trip, accuracy kept."""
    from bayesian_inference_for_nn_amd.distributions import MultivariateNormalDiagPlusLowRank
    ds = moons_dataset()
    start = model_from_json(MOONS_JSON)
    pre = SGD()
    pre.compile(HyperParameters(lr=0.1, frequency=1), MOONS_JSON, ds, verbose=False, starting_model=start, seed=2)
    pre.train(600)
    opt = SWAG()
    opt.compile(HyperParameters(lr=0.02, k=5, frequency=3, scale=1.0, batch_size=64), MOONS_JSON, ds, verbose=False,
                starting_model=pre.result()._model, seed=3)
    opt.train(60)
    assert opt._n == 60 and opt._n_cols == 5
    bm = opt.result()
    assert len(bm._distributions) == 2 and all(isinstance(d, MultivariateNormalDiagPlusLowRank) for d in bm._distributions)
    d0 = bm._distributions[0]
    assert d0._D.shape == (2 * 16 + 16, 5) and np.all(d0._diag > -1e-6)
    xt, yt = next(iter(ds.test_data.batch(ds.test_size)))
    _, mean = bm.predict(xt, nb_samples=20)
    acc = float(np.mean(np.argmax(mean, axis=1) == yt.numpy().ravel()))
    _, mean0 = pre.result().predict(xt, nb_samples=1)
    acc0 = float(np.mean(np.argmax(mean0, axis=1) == yt.numpy().ravel()))
    assert acc > 0.7 and acc >= acc0 - 0.08            # SWAG keeps the accuracy of its starting point
    bm.store(str(tmp_path / "swag"))
    back = BayesianModel.load(str(tmp_path / "swag"))
    assert [type(d).__name__ for d in back._distributions] == ["MultivariateNormalDiagPlusLowRank"] * 2
    np.testing.assert_allclose(back._distributions[1]._D, bm._distributions[1]._D, rtol=1e-6)


def test_sgd_quiet_train_is_the_step_loop():
    """verbose=False runs the whole SGD train loop on the device (pyz_sgd_run, graph replay, cut at the last
    `frequency` hit): same weights, same "mean", same epoch bookkeeping as the per-step loop."""
    def make():
        ds = moons_dataset(seed=5)
        start = model_from_json(MOONS_JSON)
        start.reset_glorot(np.random.default_rng(9))
        opt = SGD()
        return opt, ds, start
    n_it = 75                                               # 400 training rows, batch 64: 7 batches per epoch
    a, ds, start = make()
    a.compile(HyperParameters(lr=0.05, frequency=4, batch_size=64), MOONS_JSON, ds, verbose=False, starting_model=start, seed=11)
    a.train(n_it)
    b, ds2, start2 = make()
    b.compile(HyperParameters(lr=0.05, frequency=4, batch_size=64), MOONS_JSON, ds2, verbose=False, starting_model=start2, seed=11)
    for _ in range(n_it):
        last = b.step()
    assert a._n == b._n == n_it and a._epoch_num == b._epoch_num and a._seen_batches == b._seen_batches
    np.testing.assert_allclose(a._theta.cpu().numpy(), b._theta.cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(a._mean_dev.cpu().numpy(), b._mean_dev.cpu().numpy(), rtol=0, atol=1e-6)
    assert not torch.equal(a._mean_dev, a._theta)            # 75 steps: the last "mean <- weights" was at count 72
    np.testing.assert_allclose(float(a._running_dev), float(b._running_dev), rtol=1e-5)
    assert a.last_losses.shape == (n_it,) and abs(float(a._loss_dev) - float(b._loss_dev)) < 1e-6


def test_quiet_train_calls_share_their_device_buffers():
    """Successive verbose=False runs reuse one row-index table and one loss buffer (so the captured graph
    is replayed, not rebuilt); a longer run grows them.  Results equal the per-step loop throughout."""
    def make():
        ds = moons_dataset(seed=7)
        start = model_from_json(MOONS_JSON)
        start.reset_glorot(np.random.default_rng(13))
        opt = SGD()
        opt.compile(HyperParameters(lr=0.05, frequency=3, batch_size=64), MOONS_JSON, ds, verbose=False, starting_model=start, seed=14)
        return opt
    a, b = make(), make()
    a.train(20)
    first = a.last_losses.cpu().numpy().copy()
    buf = a._res_idx.data_ptr()
    a.train(33)
    assert a._res_idx.data_ptr() == buf and a._res_cap == 256
    assert first.shape == (20,) and a.last_losses.shape == (33,)
    a.train(280)                                             # beyond the first capacity: new buffers, new graph
    assert a._res_cap == 512 and a._res_idx.data_ptr() != buf
    for _ in range(333):
        b.step()
    assert a._n == b._n == 333 and a._epoch_num == b._epoch_num
    np.testing.assert_allclose(a._theta.cpu().numpy(), b._theta.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(a._mean_dev.cpu().numpy(), b._mean_dev.cpu().numpy(), rtol=0, atol=2e-6)


def test_swag_quiet_train_is_the_step_loop():
    """verbose=False runs the SWAG train loop on the device (pyz_swag_run): same weights, moments and
    deviation rows as the per-step loop, including the column that is replaced once k exist."""
    def make():
        ds = moons_dataset(seed=6)
        start = model_from_json(MOONS_JSON)
        start.reset_glorot(np.random.default_rng(10))
        opt = SWAG()
        opt.compile(HyperParameters(lr=0.03, k=4, frequency=5, scale=1.0, batch_size=64), MOONS_JSON, ds, verbose=False,
                    starting_model=start, seed=12)
        return opt
    a, b = make(), make()
    a._resident_chunks = (16, 0.75, 16)                          # several plan/launch chunks per run
    a.train(40)
    a.train(37)                                              # a second run continues the count (77 steps: 16 hits)
    for _ in range(77):
        b.step()
    assert a._n == b._n == 77 and a._n_cols == b._n_cols == 4
    for name in ("_theta", "_mean_dev", "_sq_mean_dev", "_dev_rows"):
        np.testing.assert_allclose(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy(), rtol=0, atol=1e-6, err_msg=name)


def test_predict_draws_normal_posteriors_on_the_device():
    """BayesianModel.predict with Normal posteriors: the weight draws come from the device Philox stream --
    right moments, reproducible under tfd.seed, different from draw to draw; a Deterministic layer is copied."""
    from bayesian_inference_for_nn_amd.distributions import tfd
    from bayesian_inference_for_nn_amd.distributions.tf import TensorflowProbabilityDistribution
    cfg = sequential_json(6, [40, 3], ["relu", "softmax"])
    bm = BayesianModel(cfg)
    rng = np.random.default_rng(1)
    sl0, sl1 = bm._model.spec.layer_slices()
    loc0 = (rng.normal(size=sl0.stop - sl0.start) * 0.3).astype(np.float32)
    sc0 = np.full_like(loc0, 0.05)
    det = (rng.normal(size=sl1.stop - sl1.start) * 0.3).astype(np.float32)
    bm.apply_distribution(TensorflowProbabilityDistribution(tfd.Normal(loc0, sc0)), 0, 0)
    bm.apply_distribution(TensorflowProbabilityDistribution(tfd.Deterministic(det)), 1, 1)
    tfd.seed(7)
    W = bm.sample_weights_device(4000).cpu().numpy()
    np.testing.assert_array_equal(W[:, sl1], np.repeat(det[None, :], 4000, axis=0))
    np.testing.assert_allclose(W[:, sl0].mean(0), loc0, atol=0.005)
    np.testing.assert_allclose(W[:, sl0].std(0), sc0, rtol=0.08)
    assert not np.array_equal(W[0, sl0], W[1, sl0])
    tfd.seed(7)
    np.testing.assert_array_equal(bm.sample_weights_device(4000).cpu().numpy(), W)
    x = rng.normal(size=(50, 6)).astype(np.float32)
    tfd.seed(7)
    W16 = bm.sample_weights_device(16).cpu().numpy()
    tfd.seed(7)
    samples, mean = bm.predict(x, nb_samples=16)          # the same 16 draws
    rs, rm = o_predict.predict(W16, x, o_mlp.MLPSpec((6, 40, 3), ("relu", "softmax"), "scce"))
    np.testing.assert_allclose(np.stack(samples), rs, atol=1e-5)
    np.testing.assert_allclose(mean, rm, atol=1e-5)
