"""Abstract base of the inference methods (mirrors Pyesian/optimizers/Optimizer.py:14-165:
compile-once guard, shuffled/batched training iterator, ``train`` loop with progress bar,
optional loss file and periodic ``result().store``).  The data set is uploaded to HBM once at
compile time; a batch is a slice of a per-epoch device permutation, gathered inside the kernels."""

from __future__ import annotations

import math
import os
import shutil
from abc import ABC, abstractmethod

import numpy as np

from ..losses import loss_kind
from ..nn.model import Array


class DeviceScalar:
    """A loss that stays on the device until somebody looks at it (``Optimizer.train`` prints the
    loss every step when verbose: Optimizer.py:123 -- formatting is what triggers the sync)."""

    def __init__(self, tensor, index=0, scale=1.0):
        self._t, self._i, self._s = tensor, index, scale

    def __float__(self):
        return float(self._t.reshape(-1)[self._i].item()) * self._s

    def item(self):
        return float(self)

    def numpy(self):
        return Array(np.float32(float(self)))

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    __str__ = __repr__


class Optimizer(ABC):
    def __init__(self):
        self._model_config = None
        self._hyperparameters = None
        self.__compiled = False
        self._dataset = None

    # ------------------------------------------------------------------ reference surface
    @abstractmethod
    def step(self, save_document_path=None):
        pass

    def compile(self, hyperparameters, model_config: str, dataset, verbose=True, **kwargs):
        if self.__compiled:
            raise Exception("Model Already compiled")
        else:
            self.__compiled = True
            self._hyperparameters = hyperparameters
            self._model_config = model_config
            self._dataset = dataset
            self._verbose = verbose
        self.compile_extra_components(**kwargs)

    @abstractmethod
    def compile_extra_components(self, **kwargs):
        pass

    @abstractmethod
    def update_parameters_step(self):
        pass

    @abstractmethod
    def result(self):
        pass

    def _empty_folder(self, path):
        from .._fs import empty_folder
        empty_folder(path)

    def train_with_weights_and_biases(self, nb_iterations, project_name, weights_and_biases_config):
        raise RuntimeError("wandb is not available in this environment (no network)")

    def train(self, nb_iterations: int, loss_save_document_path: str = None, model_save_frequency: int = None,
              model_save_path: str = None, weights_and_biases_log=False):
        if model_save_frequency == None and model_save_path != None:
            raise Exception("Error: save path precised and save frequency is None, please provide a savong frequency")
        if model_save_frequency != None and model_save_path == None:
            raise Exception("Error: save frequency precised and save path is None, please provide a saving path")
        if weights_and_biases_log:
            raise RuntimeError("wandb is not available in this environment (no network)")
        if loss_save_document_path != None and os.path.exists(loss_save_document_path):
            os.remove(loss_save_document_path)
        if model_save_path != None:
            self._empty_folder(model_save_path)

        quiet = (not self._verbose and loss_save_document_path is None and model_save_frequency is None)
        if quiet and self._train_resident(nb_iterations):
            return
        saved_model_nbr = 0
        for i in range(nb_iterations):
            loss = self.step(loss_save_document_path)
            self._print_progress(i / nb_iterations, loss=loss)
            if model_save_frequency != None and i % model_save_frequency == 0:
                bayesian_model = self.result()
                target = os.path.join(model_save_path, "model" + str(saved_model_nbr))
                if os.path.exists(target):
                    shutil.rmtree(target)
                os.makedirs(target)
                (bayesian_model[0] if isinstance(bayesian_model, tuple) else bayesian_model).store(target)
                saved_model_nbr += 1
        if self._verbose:
            print()

    def _train_resident(self, nb_iterations: int) -> bool:
        """Hook: run all iterations without per-step host work (returns False if unsupported)."""
        return False

    def _print_progress(self, progress: float, bar_length=10, suffix="Training", **kwargs):
        """One carriage-returned status line: "<suffix> <pct> % [====>     ] key: value ..." (the text the
        reference shows, Optimizer.py:149-160)."""
        if not self._verbose:
            return
        filled = math.ceil(progress * bar_length)
        bar = "[" + "=" * filled + (">" if filled < bar_length else "") + "]"
        fields = " ".join(f"{k}: {v}" for k, v in kwargs.items())
        print(f"\r{suffix} {math.ceil(progress * 100)} % {bar} {fields}", end="")

    def _new_progress_line(self):
        if self._verbose:
            print()

    # ------------------------------------------------------------------ device-resident data
    def _setup_backend(self, seed=None, max_particles=1, full_batch=False, chain_per_rank=True):
        """Builds the model, the kernel plan and the device copy of the training split.
        Several ranks (one process per GPU): every rank starts from the SAME base seed (the given one, else rank
        0's entropy); independent chains (chain_per_rank) use base + rank, sharded SVGD the base itself -- its
        ranks must draw one batch permutation and one particle initialisation."""
        import torch
        from .. import parallel
        from ..engine import MLPPlan, MLPSpec
        from ..nn.model import model_from_json
        self._rank, self._world = parallel.world_info()
        self._seed = parallel.shared_seed(seed) + (self._rank if chain_per_rank else 0)
        self._rng = np.random.default_rng(self._seed)
        self._net = model_from_json(self._model_config)
        self._net.reset_glorot(self._rng)
        self._loss_kind = loss_kind(self._dataset._loss)
        self._spec = MLPSpec(self._net.dims, self._net.acts, self._loss_kind)
        x, y = self._dataset.training_dataset().as_numpy()
        self._training_dataset_cardinality = len(x)
        self._x_dev = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(len(x), -1))).cuda()
        self._y_dev = self._labels_to_device(y)
        if full_batch:
            self._batch_size = len(x)
        self._plan = MLPPlan(self._spec, max_batch=max(1, min(int(self._batch_size), len(x))),
                             max_particles=max_particles)
        self._D = self._plan.D

    def _labels_to_device(self, y):
        import torch
        y = np.asarray(y)
        if self._loss_kind == "scce":
            return torch.as_tensor(np.ascontiguousarray(y.reshape(-1).astype(np.int32))).cuda()
        return torch.as_tensor(np.ascontiguousarray(y.astype(np.float32).reshape(len(y), -1))).cuda()

    def _dataset_setup(self):
        """shuffle(cardinality).batch(batch_size) + iter (Optimizer.py:35-41): a device permutation
        per epoch, ragged last batch, fresh permutation when exhausted."""
        self._training_dataset = self._dataset.training_dataset()
        self._perm_host = None
        self._perm_dev = None
        self._pos = 0
        self._epoch = -1

    def _new_epoch(self):
        self._epoch += 1
        self._perm_host = self._rng.permutation(self._training_dataset_cardinality).astype(np.int32)
        self._perm_dev = None                        # uploaded when a step() asks for a batch of it
        self._pos = 0

    def _next_batch(self):
        """(row-index view on the device, batch size, True when a new epoch started)."""
        import torch
        new_epoch = False
        if self._perm_host is None or self._pos >= self._training_dataset_cardinality:
            new_epoch = self._perm_host is not None
            self._new_epoch()
        if self._perm_dev is None:
            self._perm_dev = torch.as_tensor(self._perm_host).cuda()
        b = min(int(self._batch_size), self._training_dataset_cardinality - self._pos)
        idx = self._perm_dev[self._pos:self._pos + b]
        self._pos += b
        return idx, b, new_epoch

    def _batch_plan(self, n_steps: int):
        """Row indices of the next n_steps batches as one host (n_steps, max_batch) table + sizes (whole epochs
        are laid out with one reshape each: no per-step host work)."""
        import torch
        B = self._plan.max_batch
        N = self._training_dataset_cardinality
        bsz = int(self._batch_size)
        table = np.zeros((n_steps, B), dtype=np.int32)
        sizes = []
        self._plan_epoch_starts = []                 # steps of this plan that opened a new epoch
        s = 0
        while s < n_steps:
            if self._perm_host is None or self._pos >= N:
                if self._perm_host is not None:
                    self._plan_epoch_starts.append(s)
                self._new_epoch()
            rest = self._perm_host[self._pos:]
            nb = min(-(-len(rest) // bsz), n_steps - s)          # batches taken from this epoch
            take = min(len(rest), nb * bsz)
            full, tail = divmod(take, bsz)
            if full:
                table[s:s + full, :bsz] = rest[:full * bsz].reshape(full, bsz)
            if tail:
                table[s + full, :tail] = rest[full * bsz:take]
            sizes += [bsz] * full + ([tail] if tail else [])
            self._pos += take
            s += nb
        return torch.as_tensor(table), sizes

    def _reserve_resident(self, n_steps: int):
        """Persistent device buffers for the device-resident runs: the row-index table and the per-step losses.
        Their addresses are baked into the library's captured hipGraph, so they are kept (and only grown)
        across train() calls -- fresh tensors per call would re-capture and re-instantiate the graph each time."""
        import torch
        if n_steps > getattr(self, "_res_cap", 0):
            cap = max(256, 1 << (n_steps - 1).bit_length())
            self._res_idx = torch.zeros((cap, self._plan.max_batch), dtype=torch.int32, device="cuda")
            self._res_losses = torch.zeros(cap, device="cuda")
            self._res_cap = cap

    def _resident_buffers(self, table, n_steps: int):
        """(row-index table, loss buffer) on the device with `table` in the first n_steps slots."""
        self._reserve_resident(n_steps)
        self._res_idx[:n_steps].copy_(table)
        return self._res_idx, self._res_losses

    # steps in the first chunk of a resident run, growth from chunk to chunk, largest chunk: laying out a step's batch
    # costs the host about 15 us (one permutation per epoch), the device takes 24 us for it at C2 -- a chunk may be at most
    # 1.6 times the one the device is working through, or the device waits for its plan; the first chunk is small because
    # the device idles while it is planned
    _resident_chunks = (32, 1.5, 512)

    def _run_stream(self):
        """The stream of this optimizer's device-resident runs (graph replay needs a stream of its own)."""
        import torch
        if getattr(self, "_res_stream", None) is None:
            self._res_stream = torch.cuda.Stream()
        return self._res_stream

    @staticmethod
    def _join_run(stream):
        """train() returns when its run is done, like the reference's (Optimizer.py:94-137 is a synchronous loop): the
        HOST waits for the run stream.  Letting the caller's stream wait on the device instead (a barrier packet parked on
        a second hardware queue for the length of the run) costs every step of the run 2.3 us at C2 -- 25.8 against
        23.5 us (tools/probe_train_gap.py)."""
        stream.synchronize()

    def _warn_if_diverged(self, stream=None):
        """The device counts the steps whose loss came out NaN / Inf (survey 5.3); a quiet device-resident run prints no
        loss, so this is where a diverged chain becomes visible: RuntimeWarning with the number of such steps.
        Called where the host has just joined the run (one 4-byte copy)."""
        import warnings
        import torch
        from .._lib import E_NAN, PyzError
        try:
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
                self._plan.check_finite()
        except PyzError as e:
            if e.code != E_NAN:
                raise
            warnings.warn(f"{type(self).__name__}.train: {e}", RuntimeWarning, stacklevel=3)

    def _run_resident_chunks(self, nb_iterations: int, launch):
        """A device-resident run, planned and launched in chunks: the host lays out the batches of the next chunk
        (one permutation per epoch) while the device works through the current one.
        launch(row_idx, losses, batch_sizes, s0) enqueues steps [s0, s0 + len(batch_sizes)) on the current stream;
        returns the losses of the run (a view of the persistent buffer)."""
        import torch
        first, growth, largest = self._resident_chunks
        self._reserve_resident(nb_iterations)
        main, stream = torch.cuda.current_stream(), self._run_stream()
        s0, size = 0, float(first)
        while s0 < nb_iterations:
            n = min(max(1, int(size)), nb_iterations - s0)
            size = min(size * growth, float(largest))
            table, sizes = self._batch_plan(n)
            self._res_idx[s0:s0 + n].copy_(table)
            stream.wait_stream(main)
            with torch.cuda.stream(stream):
                launch(self._res_idx, self._res_losses, sizes, s0)
            s0 += n
        self._join_run(stream)
        self._warn_if_diverged(stream)
        return self._res_losses[:nb_iterations]

    def _layer_indices(self):
        """indices (in model.layers) of the layers that own parameters"""
        return [i for i, l in enumerate(self._net.layers) if len(l.trainable_variables) != 0]
