"""Stochastic Gradient Langevin Dynamics (mirrors Pyesian/optimizers/SGLD.py:14-166).
Hyperparameters: batch_size, lr_upper, lr_lower, lr_gamma.  One fused device step: forward,
loss, backward and the noise + parameter + running-moment update (three kernel launches)."""

import numpy as np

from ..distributions import tfd
from ..distributions.tf import TensorflowProbabilityDistribution
from ..nn import BayesianModel
from .Optimizer import DeviceScalar, Optimizer


class SGLD(Optimizer):
    def __init__(self):
        super().__init__()
        self._n = None
        self._running_loss = None
        self._lr_upper = None
        self._lr_lower = None
        self._lr_gamma = None
        self._lr = None

    def _init_sgld_lr(self):
        """SGLD.py:112-118: lr(step) = a (b + step)^-gamma, lr(0) = lr_upper, lr(n) = lr_lower."""
        n = self._nb_iterations
        l_g = np.power(self._lr_lower, 1.0 / self._lr_gamma)
        u_g = np.power(self._lr_upper, 1.0 / self._lr_gamma)
        b = -(n * l_g) / (l_g - u_g)
        a = self._lr_upper * np.power(b, self._lr_gamma)
        self._lr = lambda step: a * np.power((b + step), -self._lr_gamma)

    def compile_extra_components(self, **kwargs):
        import torch
        self._batch_size = int(self._hyperparameters.batch_size)
        self._lr_upper = self._hyperparameters.lr_upper
        self._lr_lower = self._hyperparameters.lr_lower
        self._lr_gamma = self._hyperparameters.lr_gamma
        self._setup_backend(seed=kwargs.get("seed"))
        self._merge_ranks = bool(kwargs.get("merge_ranks", True))
        self._base_model = self._net
        self._dataset_setup()
        self._theta = torch.as_tensor(self._net.weights_flat.copy()).cuda()
        self._mean_dev = torch.zeros(self._D, device="cuda")       # SGLD.py:97-110
        self._sq_mean_dev = torch.zeros(self._D, device="cuda")
        self._loss_dev = torch.zeros(1, device="cuda")
        self._running_dev = torch.zeros(1, device="cuda")
        self._weight_layers_indices = self._layer_indices()
        self._n = 0
        self._running_loss = 0

    def step(self, save_document_path=None):
        idx, b, _ = self._next_batch()
        lr = float(self._lr(self._n))
        self._plan.sgld_step(self._theta, self._mean_dev, self._sq_mean_dev, self._x_dev, self._y_dev, lr, self._n,
                             self._seed, self._loss_dev, batch=b, row_idx=idx)
        self._running_dev += self._loss_dev                        # SGLD.py:58
        if save_document_path != None:
            with open(save_document_path, "a") as losses_file:
                losses_file.write(str(float(self._loss_dev.item())))
        self._n += 1
        return DeviceScalar(self._running_dev.clone(), 0, 1.0 / self._n)

    def train(self, nb_iterations: int, loss_save_document_path: str = None, model_save_frequency: int = None,
              model_save_path: str = None, weights_and_biases_log=False):
        self._nb_iterations = nb_iterations
        self._init_sgld_lr()
        super().train(nb_iterations, loss_save_document_path, model_save_frequency, model_save_path,
                      weights_and_biases_log)

    def _train_resident(self, nb_iterations: int) -> bool:
        """verbose=False: all steps in one device-resident run (hipGraph replay, no host sync)."""
        import torch
        lrs = np.asarray(self._lr(self._n + np.arange(nb_iterations, dtype=np.float64))).astype(np.float32).tolist()

        def launch(idx, loss_buf, sizes, s0):
            self._plan.sgld_run(self._theta, self._mean_dev, self._sq_mean_dev, self._x_dev, self._y_dev, idx, sizes,
                                lrs[s0:s0 + len(sizes)], self._n + s0, self._seed, loss_buf, use_graph=True, slot0=s0)
        losses = self._run_resident_chunks(nb_iterations, launch)
        self._running_dev += losses.sum()
        self._n += nb_iterations
        self.last_losses = losses.clone()          # the buffer itself is reused by the next run
        return True

    def update_parameters_step(self):
        return super().update_parameters_step()

    def result(self) -> BayesianModel:
        model = BayesianModel(self._model_config)
        mean_dev, sq_dev = self._mean_dev, self._sq_mean_dev
        if self._world > 1 and self._merge_ranks:
            # one chain per rank (no data-path collective): the running moments pool with the chains' step counts
            from .. import parallel
            mean_dev, sq_dev, self.pooled_steps = parallel.merge_moment_chains(mean_dev, sq_dev, self._n)
        mean = mean_dev.cpu().numpy()
        sq_mean = sq_dev.cpu().numpy()
        for sl, layer_idx in zip(self._spec.layer_slices(), self._weight_layers_indices):
            # Normal(loc = mean, scale = sq_mean - mean^2): the variance is used as the scale, as written (SGLD.py:151-154)
            dist = TensorflowProbabilityDistribution(tfd.Normal(mean[sl].copy(), (sq_mean[sl] - mean[sl] ** 2).copy()))
            model.apply_distribution(dist, layer_idx, layer_idx)
        model._model.set_flat(self._theta.cpu().numpy())
        return model
