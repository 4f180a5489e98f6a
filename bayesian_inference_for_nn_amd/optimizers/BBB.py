"""Bayes-by-Backprop (mirrors Pyesian/optimizers/BBB.py:12-326).  Hyperparameters: batch_size,
lr, alpha, [pi]; kwargs prior, [prior2].  Device step: sample w + log-likelihood sums, fused
forward/backward, closed-form mu/rho update (BBB.py:152-201)."""

import math

import numpy as np

from ..distributions import GaussianPrior, tfd
from ..distributions.tf import TensorflowProbabilityDistribution
from ..nn import BayesianModel
from .Optimizer import DeviceScalar, Optimizer


class ResultTuple(tuple):
    """``BBB.result()`` returns ``(model, train_losses, val_losses)`` (BBB.py:323) while several
    reference drivers call ``.predict`` / ``.store`` on the return value (BBB_mnist.py:53,59):
    unpacks as three items and forwards attribute access to the model."""

    def __getattr__(self, name):
        return getattr(self[0], name)


def softplus(x):
    return np.logaddexp(0.0, x)


class BBB(Optimizer):
    def __init__(self):
        super().__init__()
        self._lr = None
        self._alpha = None
        self._prior2 = None
        self._prior = None
        self._step = 0
        self.val_losses = []
        self.train_losses = []

    def compile_extra_components(self, **kwargs):
        import torch
        from ..engine import MLPPlan
        self._prior = kwargs["prior"]
        self._prior2 = kwargs["prior2"] if "prior2" in kwargs else GaussianPrior(0.0, 0.0)
        self._lr = self._hyperparameters.lr
        self._pi = getattr(self._hyperparameters, 'pi', None) if hasattr(self._hyperparameters, 'pi') else 1
        self._batch_size = int(self._hyperparameters.batch_size)
        if isinstance(self._prior._mean, int) or isinstance(self._prior._mean, float):      # BBB.py:265-270
            sign = self._prior._std_dev / abs(self._prior._std_dev)
            self._prior = GaussianPrior(
                self._prior._mean * self._pi + self._prior2._mean * (1 - self._pi),
                sign * math.sqrt((self._prior._std_dev * self._pi) ** 2 + (self._prior2._std_dev * (1 - self._pi)) ** 2),
            )
        self._alpha = self._hyperparameters.alpha
        self._setup_backend(seed=kwargs.get("seed"))
        self._base_model = self._net
        self._dataset_setup()
        self._priors_list = self._prior.get_model_priors(self._net)
        mu, rho = self._prior.flat(self._net)                   # BBB.py:277-296: posterior <- prior (raw rho)
        self._mu = torch.as_tensor(mu.copy()).cuda()
        self._rho = torch.as_tensor(rho.copy()).cuda()
        if self._prior.is_scalar():
            self._pm, self._pr, self._pm_vec, self._pr_vec = float(self._prior._mean), float(self._prior._std_dev), None, None
        else:                                                   # list-valued prior: no mixing (BBB.py:265), per-element vectors
            self._pm, self._pr = 0.0, 1.0
            self._pm_vec, self._pr_vec = self._mu.clone(), self._rho.clone()
        self._w = torch.zeros(self._D, device="cuda")
        self._cost = torch.zeros(4, device="cuda")
        self._weight_layers_indices = self._layer_indices()
        # validation split, forwarded with the sampled weights on 9 steps out of 10 (BBB.py:203-209)
        vx, vy = self._dataset.valid_data.as_numpy()
        self._val_n = len(vx)
        if self._val_n > 0:
            self._vx = torch.as_tensor(np.ascontiguousarray(np.asarray(vx, np.float32).reshape(len(vx), -1))).cuda()
            self._vy = self._labels_to_device(vy)
            self._val_plan = MLPPlan(self._spec, max_batch=self._val_n)

    def step(self, save_document_path=None):
        self._step += 1
        idx, b, _ = self._next_batch()
        self._plan.bbb_step(self._mu, self._rho, self._w, self._x_dev, self._y_dev, self._lr, self._alpha,
                            self._pm, self._pr, self._step, self._seed, self._cost, batch=b, row_idx=idx,
                            prior_mean_vec=self._pm_vec, prior_rho_vec=self._pr_vec)
        likelihood = DeviceScalar(self._cost.clone(), 0)
        if save_document_path != None:
            with open(save_document_path, "a") as losses_file:
                losses_file.write(str(float(likelihood)) + "\n")
        if self._step % 10:                                     # BBB.py:203: nine steps out of ten, as written
            if self._val_n > 0:
                vloss, _ = self._val_plan.loss_grad(self._w, self._vx, self._vy, want_grad=False)
                self.val_losses.append(DeviceScalar(vloss, 0))
            self.train_losses.append(likelihood)
        return likelihood

    def _train_resident(self, nb_iterations: int) -> bool:
        """verbose=False: all steps in device-resident runs (pyz_bbb_run: hipGraph replay, batches assembled one step
        ahead, the validation forward of BBB.py:203-209 inside the run).  Equal to the per-step loop bit for bit."""
        import torch
        from .._lib import PyzError
        if nb_iterations <= 0:
            return True
        if self._spec.dims[-1] > 32:                            # the chained step needs the fused head
            return False
        self._reserve_resident(nb_iterations)
        if getattr(self, "_res_costs", None) is None or self._res_costs.shape[0] < self._res_cap:
            self._res_costs = torch.zeros((self._res_cap, 4), device="cuda")
            self._res_val = torch.zeros(self._res_cap, device="cuda")
        lr, step0 = float(self._lr), self._step + 1
        has_val = self._val_n > 0

        def launch(idx, _losses, sizes, s0):
            self._plan.bbb_run(self._mu, self._rho, self._w, self._x_dev, self._y_dev, idx, sizes, [lr] * len(sizes),
                               self._alpha, self._pm, self._pr, step0 + s0, self._seed, self._res_costs, use_graph=True,
                               slot0=s0, prior_mean_vec=self._pm_vec, prior_rho_vec=self._pr_vec,
                               val_plan=self._val_plan if has_val else None, val_x=self._vx if has_val else None,
                               val_y=self._vy if has_val else None, val_losses_out=self._res_val if has_val else None)
        try:
            self._run_resident_chunks(nb_iterations, launch)
        except PyzError:
            if self._step + 1 != step0:
                raise
            return False
        costs, vals = self._res_costs[:nb_iterations].clone(), self._res_val[:nb_iterations].clone()
        for i in range(nb_iterations):
            if (step0 + i) % 10:                                # BBB.py:203: nine steps out of ten, as written
                if has_val:
                    self.val_losses.append(DeviceScalar(vals, i))
                self.train_losses.append(DeviceScalar(costs, 4 * i))
        self._cost[:4].copy_(costs[-1])
        self._step += nb_iterations
        return True

    def update_parameters_step(self):
        pass

    def result(self):
        model = BayesianModel(self._model_config)
        mu, rho = self._mu.cpu().numpy(), self._rho.cpu().numpy()
        for sl, layer_idx in zip(self._spec.layer_slices(), self._weight_layers_indices):
            dist = TensorflowProbabilityDistribution(tfd.Normal(mu[sl].copy(), softplus(rho[sl]).astype(np.float32)))
            model.apply_distribution(dist, layer_idx, layer_idx)
        return ResultTuple((model, self.train_losses, self.val_losses))
