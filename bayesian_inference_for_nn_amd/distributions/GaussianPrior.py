"""Gaussian prior specification (mirrors Pyesian/distributions/GaussianPrior.py:5-121)."""

import numpy as np

from . import tfd


class GaussianPrior:
    """mean / rho are both int-or-float (one value for the whole model) or both lists (one
    value per layer).  ``rho`` plays the role of the standard deviation, or of its pre-image
    for methods that transform it (BBB applies a softplus)."""

    def __init__(self, mean, rho):
        if (type(mean)) != type(rho):
            raise Exception("mean and std dev must have the same type")
        self._mean = mean
        self._std_dev = rho

    def _get_priors_from_int_or_float(self, model):
        priors_list = []
        for layer in model.layers:
            if len(layer.trainable_variables) != 0:
                priors_list.append([tfd.Normal(self._mean * np.ones(w.shape, np.float32),
                                               self._std_dev * np.ones(w.shape, np.float32))
                                    for w in layer.trainable_variables])
            else:
                priors_list.append(None)
        return priors_list

    def _get_priors_from_list(self, model):
        priors_list = []
        for layer_idx, layer in enumerate(model.layers):
            if len(layer.trainable_variables) != 0:
                priors_list.append([tfd.Normal(self._mean[layer_idx] * np.ones(w.shape, np.float32),
                                               self._std_dev[layer_idx] * np.ones(w.shape, np.float32))
                                    for w in layer.trainable_variables])
            else:
                priors_list.append(None)
        return priors_list

    def get_model_priors(self, model):
        if isinstance(self._mean, int) or isinstance(self._mean, float):
            return self._get_priors_from_int_or_float(model)
        if isinstance(self._mean, list) and (all(isinstance(m, int) for m in self._mean)
                                             or all(isinstance(m, float) for m in self._mean)):
            return self._get_priors_from_list(model)
        # the reference's per-tensor variant (GaussianPrior.py:71-98) never returns a value
        raise Exception("mean and standard deviation should be an int, a float, a list or a tensor")

    def flat(self, model):
        """(mean, rho) as flat float32 vectors in the model's parameter order."""
        mus, rhos = [], []
        for p in self.get_model_priors(model):
            if p:
                for d in p:
                    mus.append(d.mean().reshape(-1))
                    rhos.append(d.stddev().reshape(-1))
        return np.concatenate(mus), np.concatenate(rhos)

    def is_scalar(self) -> bool:
        return isinstance(self._mean, (int, float))
