import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bayesian_inference_for_nn_amd import synth, engine
from bayesian_inference_for_nn_amd.distributions import tfd
from bayesian_inference_for_nn_amd.distributions.tf import TensorflowProbabilityDistribution
from bayesian_inference_for_nn_amd.nn import BayesianModel, sequential_json
cfg = sequential_json(784, [200, 10], ["relu", "softmax"])
x, _ = synth.mnist_like(10000)
model = BayesianModel(cfg)
rng = np.random.default_rng(0)
layers = [i for i, l in enumerate(model._model.layers) if len(l.trainable_variables) != 0]
for li in layers:
    n = sum(int(np.prod(v.shape)) for v in model._model.layers[li].trainable_variables)
    model.apply_distribution(TensorflowProbabilityDistribution(tfd.Normal((rng.normal(size=n) * 0.05).astype(np.float32), np.full(n, 0.01, np.float32))), li, li)
model.predict(x, nb_samples=100); model.predict(x, nb_samples=100)
torch.cuda.synchronize()
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(3): model.predict(x, nb_samples=100)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
with engine.KernelProbe(4096) as kp:
    model.predict(x, nb_samples=100)
print(json.dumps({k: [c, round(us, 1), round(c*us/1e3, 3)] for k, (c, us) in kp.by_kernel().items()}))
