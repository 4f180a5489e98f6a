"""N1 (BASELINE.json north_star: "BBB_mnist.py / HMC_classification.py run unchanged"): the two named driver
scripts of the reference, read from /root/reference and NOT copied, under the `compat/` stand-ins:

  * every module / name they import resolves (tensorflow stand-in, Pyesian -> this package, sklearn, numpy);
  * their own `run_experiment` body runs unchanged up to `Optimizer.compile(...)` and hands it what compile
    expects -- a HyperParameters bag with the method's keys, a Keras-2.15 JSON this package's parser accepts
    with the script's architecture, this package's Dataset with the 80/10/10 split, a GaussianPrior kwarg.

compile() itself needs the MI355X (there is no CPU path): it is replaced by a probe here; the whole flow --
compile, train, result, predict, Metrics / Plotter -- is exercised on the GPU by
tests/test_gpu_surface.py::test_compat_* with a script written in the same style.  Skipped where the reference
tree is absent (the GPU box)."""

import ast
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
SCRIPTS = {"BBB_mnist.py": ("BBB", dict(lr=0.01, alpha=0.3, batch_size=64, hidden_dims=20, train_steps=3)),
           "HMC_classification.py": ("HMC", dict(epsilon=0.005, m=0.5, L=3, train_steps=2))}

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree does not travel to the GPU box")


class ReachedCompile(Exception):
    def __init__(self, optimizer, args, kwargs):
        super().__init__("compile reached")
        self.optimizer, self.args, self.kwargs = optimizer, args, kwargs


@pytest.fixture()
def compat_path(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "compat"))
    for name in [n for n in sys.modules if n == "tensorflow" or n.startswith("tensorflow.") or n == "Pyesian" or n.startswith("Pyesian.")]:
        monkeypatch.delitem(sys.modules, name)
    yield
    for name in [n for n in sys.modules if n == "tensorflow" or n.startswith("tensorflow.") or n == "Pyesian" or n.startswith("Pyesian.")]:
        sys.modules.pop(name, None)


@pytest.mark.parametrize("script", list(SCRIPTS))
def test_import_set_of_the_named_script_resolves(script, compat_path):
    tree = ast.parse(open(os.path.join(REFERENCE, script)).read())
    seen = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                importlib.import_module(a.name)
                seen += 1
        elif isinstance(node, ast.ImportFrom):
            mod = importlib.import_module(node.module)
            for a in node.names:
                assert hasattr(mod, a.name), f"{script}: from {node.module} import {a.name}"
                seen += 1
    assert seen >= 8
    import tensorflow as tf
    assert "standin" in tf.__version__            # the stand-in, not a TensorFlow that appeared on the path


@pytest.mark.parametrize("script", list(SCRIPTS))
def test_named_script_runs_unchanged_up_to_compile(script, compat_path, monkeypatch):
    import warnings
    from bayesian_inference_for_nn_amd.datasets import Dataset
    from bayesian_inference_for_nn_amd.distributions import GaussianPrior
    from bayesian_inference_for_nn_amd.nn import model_from_json
    from bayesian_inference_for_nn_amd.optimizers import Optimizer
    from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters

    def probe(self, *args, **kwargs):
        raise ReachedCompile(self, args, kwargs)

    monkeypatch.setattr(Optimizer, "compile", probe)
    path = os.path.join(REFERENCE, script)
    ns = {"__name__": "reference_script", "__file__": path}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")             # Dataset('mnist') announces its synthetic substitute
        exec(compile(open(path).read(), path, "exec"), ns)        # definitions only: the __main__ block is skipped
        method, params = SCRIPTS[script]
        with pytest.raises(ReachedCompile) as hit:
            ns["run_experiment"](**params)
    opt, args, kwargs = hit.value.optimizer, hit.value.args, hit.value.kwargs
    assert type(opt).__name__ == method
    hyp, model_config, dataset = args[:3]
    assert isinstance(hyp, HyperParameters) and isinstance(dataset, Dataset) and isinstance(kwargs["prior"], GaussianPrior)
    net = model_from_json(model_config)
    if method == "BBB":
        assert (hyp.lr, hyp.alpha, hyp.batch_size) == (0.01, 0.3, 64)
        assert net.dims == (784, 20, 10) and net.acts == ("relu", "softmax")
        assert dataset.train_size == 48_000 and dataset.test_size == 6_000 and dataset.train_data.x.shape[1:] == (28, 28)
    else:
        assert (hyp.epsilon, hyp.m, hyp.L) == (0.005, 0.5, 3)
        assert net.dims == (2, 50, 2) and net.acts == ("relu", "softmax")
        assert dataset.train_size == 1600 and kwargs["prior"]._std_dev == -1.0
