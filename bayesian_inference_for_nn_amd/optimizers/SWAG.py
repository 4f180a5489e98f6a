"""SWAG (mirrors Pyesian/optimizers/SWAG.py:13-149): SGD steps from a starting model plus running
first / second moments and a k-column deviation matrix; posterior =
MultivariateNormalDiagPlusLowRank per layer.  Hyperparameters: batch_size, lr, k, scale, frequency;
kwarg starting_model.  One fused device step (the update, the moments and the deviation row are
written in the epilogue of the weight-gradient kernel)."""

from math import sqrt

import numpy as np

from ..distributions.MultivariateNormalDiagPlusLowRank import MultivariateNormalDiagPlusLowRank
from ..nn import BayesianModel
from .Optimizer import DeviceScalar, Optimizer


class SWAG(Optimizer):
    def __init__(self):
        super().__init__()
        self._n = None
        self._lr = None
        self._frequency = None
        self._k = None

    def compile_extra_components(self, **kwargs):
        import torch
        self._k = int(self._hyperparameters.k)
        self._frequency = int(self._hyperparameters.frequency)
        self._lr = self._hyperparameters.lr
        self._scale = self._hyperparameters.scale
        self._batch_size = int(self._hyperparameters.batch_size)
        self._setup_backend(seed=kwargs.get("seed"))
        self._net.set_weights(kwargs["starting_model"].get_weights())     # SWAG.py:107-108
        self._base_model = self._net
        self._dataset_setup()
        self._theta = torch.as_tensor(self._net.weights_flat.copy()).cuda()
        self._mean_dev = torch.zeros(self._D, device="cuda")               # SWAG.py:113-127
        self._sq_mean_dev = torch.zeros(self._D, device="cuda")
        self._dev_rows = torch.zeros((self._k, self._D), device="cuda")    # row c = column c of the reference's matrix
        self._n_cols = 0
        self._loss_dev = torch.zeros(1, device="cuda")
        self._weight_layers_indices = self._layer_indices()
        self._n = 0

    def step(self, save_document_path=None):
        idx, b, _ = self._next_batch()
        update = self._n % self._frequency == 0                             # SWAG.py:72
        # columns are appended until there are k; afterwards the LAST one is replaced (SWAG.py:85-89, as written)
        col = min(self._n_cols, self._k - 1)
        self._plan.swag_step(self._theta, self._mean_dev, self._sq_mean_dev, self._dev_rows[col] if update else None,
                             self._x_dev, self._y_dev, self._lr, self._n, update, self._loss_dev, batch=b, row_idx=idx)
        if update and self._n_cols < self._k:
            self._n_cols += 1
        if save_document_path != None:
            with open(save_document_path, "a") as losses_file:
                losses_file.write(str(float(self._loss_dev.item())))
        self._n += 1
        return DeviceScalar(self._loss_dev.clone(), 0)

    def _train_resident(self, nb_iterations: int) -> bool:
        """verbose=False: all steps in one device-resident run; the moment / deviation-column bookkeeping is a
        function of the step count and runs on the device (the chain starts at count 0: compile sets it)."""
        import torch
        from .._lib import PyzError
        if nb_iterations <= 0:
            return True
        def launch(idx, loss_buf, sizes, s0):
            self._plan.swag_run(self._theta, self._mean_dev, self._sq_mean_dev, self._dev_rows, self._frequency,
                                self._x_dev, self._y_dev, idx, sizes, [float(self._lr)] * len(sizes), self._n + s0,
                                loss_buf, slot0=s0)
        try:
            losses = self._run_resident_chunks(nb_iterations, launch)
        except PyzError:                              # shapes the fused step does not take: per-step loop
            return False
        self._n += nb_iterations
        self._n_cols = min(self._k, -(-self._n // self._frequency))      # hits among counts 0 .. n - 1
        self._loss_dev.copy_(losses[-1:])
        self.last_losses = losses.clone()          # the buffer itself is reused by the next run
        return True

    def update_parameters_step(self):
        pass

    def result(self) -> BayesianModel:
        model = BayesianModel(self._model_config)
        mean, sq_mean = self._mean_dev.cpu().numpy(), self._sq_mean_dev.cpu().numpy()
        dev = self._dev_rows[:self._n_cols].cpu().numpy()
        for sl, layer_idx in zip(self._spec.layer_slices(), self._weight_layers_indices):
            dist = MultivariateNormalDiagPlusLowRank(mean[sl].copy(), (sq_mean[sl] - mean[sl] ** 2).copy(),
                                                     sqrt(self._scale / (self._k - 1)) * dev[:, sl].T)
            model.apply_distribution(dist, layer_idx, layer_idx)
        model._model.set_flat(self._theta.cpu().numpy())
        return model
