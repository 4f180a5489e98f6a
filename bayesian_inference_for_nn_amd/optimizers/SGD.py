"""Plain mini-batch SGD; the "posterior" is Deterministic(last weights) per layer
(mirrors Pyesian/optimizers/SGD.py:13-149).  Hyperparameters: lr, frequency, batch_size;
kwarg starting_model (its weights are copied, SGD.py:121-122)."""

import numpy as np

from ..distributions import tfd
from ..distributions.tf import TensorflowProbabilityDistribution
from ..nn import BayesianModel
from .Optimizer import DeviceScalar, Optimizer


class SGD(Optimizer):
    def __init__(self):
        super().__init__()
        self._n = None
        self._lr = None
        self._frequency = None
        self._k = None
        self._running_loss = 0
        self._seen_batches = 0
        self._epoch_num = 1

    def compile_extra_components(self, **kwargs):
        import torch
        self._frequency = self._hyperparameters.frequency
        self._lr = self._hyperparameters.lr
        self._batch_size = self._hyperparameters.batch_size
        self._setup_backend(seed=kwargs.get("seed"))
        start = kwargs["starting_model"]                       # KeyError if absent, like the reference
        self._net.set_weights(start.get_weights())
        self._base_model = self._net
        self._dataset_setup()
        self._theta = torch.as_tensor(self._net.weights_flat.copy()).cuda()
        self._mean_dev = self._theta.clone()                   # SGD.py:100-107: the "mean" starts as the weights
        self._loss_dev = torch.zeros(1, device="cuda")
        self._running_dev = torch.zeros(1, device="cuda")
        self._weight_layers_indices = self._layer_indices()
        self._n = 0

    def step(self, save_document_path=None):
        idx, b, new_epoch = self._next_batch()
        self._seen_batches += 1                                # SGD.py:45
        if new_epoch:                                          # SGD.py:48-54
            self._seen_batches = 1
            self._running_dev.zero_()
            self._epoch_num += 1
        self._plan.sgd_step(self._theta, self._x_dev, self._y_dev, self._lr, self._loss_dev, batch=b, row_idx=idx)
        self._running_dev += self._loss_dev                    # SGD.py:60
        if save_document_path != None:
            with open(save_document_path, "a") as losses_file:
                losses_file.write(str(float(self._loss_dev.item())))
        if self._n % self._frequency == 0:                     # SGD.py:78-84: mean <- theta
            self._mean_dev.copy_(self._theta)
        self._n += 1
        return DeviceScalar(self._running_dev.clone(), 0, 1.0 / self._seen_batches)

    def _train_resident(self, nb_iterations: int) -> bool:
        """verbose=False: all steps in device-resident runs (hipGraph replay, no per-step host work).  The
        "mean" of SGD.py:78-84 is the weights after the last step whose count is a multiple of `frequency`:
        the run is cut there, the weights copied, and the rest follows."""
        import torch
        from .._lib import PyzError
        if nb_iterations <= 0:
            return True
        table, sizes = self._batch_plan(nb_iterations)
        idx, loss_buf = self._resident_buffers(table, nb_iterations)
        losses = loss_buf[:nb_iterations]
        lrs = [float(self._lr)] * nb_iterations
        freq = int(self._frequency)
        hits = [s for s in range(nb_iterations) if (self._n + s) % freq == 0]
        cut = hits[-1] + 1 if hits else 0             # steps [0, cut) end with the last "mean <- weights"
        stream = self._run_stream()
        stream.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(stream):
                if cut > 0:
                    self._plan.sgd_run(self._theta, self._x_dev, self._y_dev, idx, sizes[:cut], lrs[:cut], losses)
                    self._mean_dev.copy_(self._theta)
                if cut < nb_iterations:
                    self._plan.sgd_run(self._theta, self._x_dev, self._y_dev, idx, sizes[cut:], lrs[cut:], losses, slot0=cut)
        except PyzError:                              # shapes the fused step does not take: per-step loop
            if cut > 0:
                raise
            return False
        self._join_run(stream)
        self._warn_if_diverged(stream)
        # epoch bookkeeping of step() (SGD.py:45-60) for the steps just run
        last_epoch = self._plan_epoch_starts[-1] if self._plan_epoch_starts else None
        if last_epoch is None:
            self._seen_batches += nb_iterations
            self._running_dev += losses.sum()
        else:
            self._epoch_num += len(self._plan_epoch_starts)
            self._seen_batches = nb_iterations - last_epoch
            self._running_dev.copy_(losses[last_epoch:].sum().reshape(1))
        self._loss_dev.copy_(losses[-1:])
        self._n += nb_iterations
        self.last_losses = losses.clone()          # the buffer itself is reused by the next run
        return True

    def result(self) -> BayesianModel:
        model = BayesianModel(self._model_config)
        mean = self._mean_dev.cpu().numpy()
        for sl, layer_idx in zip(self._spec.layer_slices(), self._weight_layers_indices):
            dist = TensorflowProbabilityDistribution(tfd.Deterministic(mean[sl].copy()))
            model.apply_distribution(dist, layer_idx, layer_idx)
        model._model.set_flat(self._theta.cpu().numpy())
        return model

    def update_parameters_step(self):
        pass
