"""Result container of every optimizer: a model + per-layer-interval weight distributions,
Monte-Carlo ``predict``, directory ``store`` / ``load``
(mirrors Pyesian/nn/BayesianModel.py:10-203; the read-out is batched on the GPU:
all ``nb_samples`` weight draws go through one particle-batched forward, pyz_predict)."""

from __future__ import annotations

import json
import os
import shutil

import numpy as np

from .model import Array, DenseNet, model_from_json


class BayesianModel:
    def __init__(self, model_config: str):
        self._model_config = model_config
        model: DenseNet = model_from_json(model_config)
        self._n_layers = len(model.layers)
        self._layers_dtbn_intervals = []
        self._distributions = []
        self._model = model
        self._plan = None

    # ------------------------------------------------------------------ distributions
    def apply_distribution(self, distribution, start_layer: int, end_layer: int):
        """Insertion rule of BayesianModel.py:25-48, as written: the first interval is appended;
        a later one is inserted after the first stored interval whose start is smaller; an
        interval that is not greater than any stored start is silently dropped."""
        if start_layer > end_layer:
            raise ValueError('starting_layer must be less than end_layer')
        elif start_layer < 0 or end_layer >= self._n_layers:
            raise ValueError('out of bounds')
        interval = [start_layer, end_layer]
        if len(self._layers_dtbn_intervals) == 0:
            self._layers_dtbn_intervals.append(interval)
            self._distributions.append(distribution)
            return
        for i in range(len(self._layers_dtbn_intervals)):
            if start_layer > self._layers_dtbn_intervals[i][0]:
                self._layers_dtbn_intervals = self._layers_dtbn_intervals[:i + 1] + [interval] + \
                    self._layers_dtbn_intervals[i + 1:]
                self._distributions = self._distributions[:i + 1] + [distribution] + self._distributions[i + 1:]
                break

    def apply_distributions_layers(self, layer_list, dtbn_list):
        self._layers_dtbn_intervals = layer_list
        self._distributions = dtbn_list

    # ------------------------------------------------------------------ sampling
    def _interval_slices(self):
        """flat slice of the parameters covered by each interval (layers start..end inclusive)."""
        lay_slices = {}
        for layer_idx, layer in enumerate(self._model.layers):
            if layer._dense_index is not None:
                lay_slices[layer_idx] = self._model.spec.layer_slices()[layer._dense_index]
        out = []
        for start, end in self._layers_dtbn_intervals:
            idxs = [i for i in range(start, end + 1) if i in lay_slices]
            if idxs:
                out.append(slice(lay_slices[idxs[0]].start, lay_slices[idxs[-1]].stop))
            else:
                out.append(slice(0, 0))
        return out

    def sample_weights_matrix(self, n: int) -> np.ndarray:
        """(n, D) float32: n joint draws, one ``Distribution`` draw per interval (BayesianModel.py:63-77)."""
        W = np.repeat(self._model.weights_flat[None, :], n, axis=0).astype(np.float32)
        for dist, sl in zip(self._distributions, self._interval_slices()):
            if sl.stop > sl.start:
                draws = np.asarray(dist.sample_n(n), dtype=np.float32)
                W[:, sl] = draws[:, : sl.stop - sl.start]
        return W

    def sample_weights_device(self, n: int):
        """The same n joint draws as a CUDA tensor.  Normal / Deterministic posteriors are drawn on the device
        (Philox: 100 draws of a 159 010-parameter posterior cost 0.2 s of host random numbers otherwise); the
        other distributions are drawn on the host and uploaded."""
        import torch
        slices = self._interval_slices()
        capable = [getattr(d, "sample_n_device", None) is not None and getattr(d, "_size", 0) <= sl.stop - sl.start
                   for d, sl in zip(self._distributions, slices)]
        if not any(capable):
            return torch.as_tensor(self.sample_weights_matrix(n)).cuda()
        W = torch.as_tensor(self._model.weights_flat.astype(np.float32)).cuda().repeat(n, 1).contiguous()
        for dist, sl, cap in zip(self._distributions, slices, capable):
            if sl.stop <= sl.start:
                continue
            if cap and dist.sample_n_device(n, W, sl.start):
                continue
            draws = np.asarray(dist.sample_n(n), dtype=np.float32)
            W[:, sl] = torch.as_tensor(np.ascontiguousarray(draws[:, : sl.stop - sl.start])).cuda()
        return W

    def _sample_weights(self):
        self._model.set_flat(self.sample_weights_matrix(1)[0])

    def sample_model(self) -> DenseNet:
        self._sample_weights()
        model = model_from_json(self._model_config)
        model.set_flat(self._model.weights_flat)
        return model

    def sample_n_models(self, n) -> list:
        return [self.sample_model() for _ in range(n)]

    # ------------------------------------------------------------------ read-out
    _predict_rows_cap = 16384      # rows per launch sequence of predict (more rows: several sequences, results joined on the device)

    def predict(self, x, nb_samples: int, y_true=None, loss_func=None):
        """(list of per-sample outputs, their mean); NaN outputs count as 0 (BayesianModel.py:106-129)."""
        import torch
        from ..engine import MLPPlan
        x = np.asarray(x.numpy() if hasattr(x, "numpy") else x)
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(len(x), -1))   # (no copy of float32 input)
        nb_samples = int(nb_samples)
        Wd = self.sample_weights_device(nb_samples)
        n = len(x)
        # bound the activation workspace: rows x samples per launch (the plan keeps an activation and a delta buffer per
        # layer: 2 * sum(widths) floats per (sample, row)).  2^29 floats = 2 GiB of the 288: 100 draws x 10 000 rows of the
        # 784 -> 200 -> 10 model go out as ONE launch per layer (3.35 ms for the wide layer against 7 x 0.61 ms in seven)
        rows = min(n, int(self._predict_rows_cap))
        per = 2 * sum(int(d) for d in self._model.dims[1:])
        chunk_s = max(1, min(nb_samples, int(os.environ.get("PYZ_PREDICT_WS", 1 << 29)) // max(1, rows * per)))
        if self._plan is None or self._plan.max_batch < rows or self._plan.max_particles < chunk_s:
            self._plan = MLPPlan(self._model.spec, max_batch=rows, max_particles=chunk_s)
        xd = torch.as_tensor(x).cuda()
        C_out = int(self._model.dims[-1])
        full = mean_full = None
        for r0 in range(0, n, rows):
            samples_d, mean_d = self._plan.predict(Wd, xd[r0:r0 + rows].contiguous())
            if rows >= n:
                full, mean_full = samples_d, mean_d
            else:
                if full is None:
                    full = torch.empty((nb_samples, n, C_out), dtype=torch.float32, device=xd.device)
                    mean_full = torch.empty((n, C_out), dtype=torch.float32, device=xd.device)
                full[:, r0:r0 + rows] = samples_d
                mean_full[r0:r0 + rows] = mean_d
        # one device-to-host copy of each result into pinned memory (the 40 MB sample tensor of 100 draws x 10 000 rows
        # moves at PCIe rate instead of through the driver's pageable staging); the NumPy views keep the buffers alive
        samples_h = torch.empty(full.shape, dtype=torch.float32, pin_memory=True)
        mean_h = torch.empty(mean_full.shape, dtype=torch.float32, pin_memory=True)
        samples_h.copy_(full, non_blocking=True)
        mean_h.copy_(mean_full, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        samples = samples_h.numpy()
        mean = mean_h.numpy()
        self._model.set_flat(Wd[-1].cpu().numpy())      # the reference leaves the last draw assigned
        return [Array(s) for s in samples], Array(mean)

    # ------------------------------------------------------------------ persistence
    @classmethod
    def load(cls, model_path: str, custom_distribution_register=None) -> "BayesianModel":
        from ..distributions import MultivariateNormalDiagPlusLowRank, Sampled
        from ..distributions.tf import TensorflowProbabilityDistribution
        register = {"Sampled": Sampled, "TensorflowProbabilityDistribution": TensorflowProbabilityDistribution,
                    "MultivariateNormalDiagPlusLowRank": MultivariateNormalDiagPlusLowRank}
        register.update(custom_distribution_register or {})
        with open(os.path.join(model_path, "config.json"), "r") as f:
            bayesian_model = BayesianModel(f.read())
        layers_intervals = []
        with open(os.path.join(model_path, "layers_config.txt"), "r") as f:
            n_intervals = int(f.readline())
            for _ in range(n_intervals):
                layers_intervals.append((f.readline()[:-1], int(f.readline()), int(f.readline())))
        for i, (name, start, end) in enumerate(layers_intervals):
            dist = register[name].load(os.path.join(model_path, "distribution" + str(i)))
            bayesian_model.apply_distribution(dist, start, end)
        return bayesian_model

    def _empty_folder(self, path):
        from .._fs import empty_folder
        empty_folder(path)

    def store(self, model_path: str):
        """Directory format of BayesianModel.py:177-203: config.json, layers_config.txt
        (count, then ClassName / start / end per interval), distribution<i>/."""
        if not os.path.exists(model_path):
            os.makedirs(model_path)
        self._empty_folder(model_path)
        with open(os.path.join(model_path, "config.json"), "w") as f:
            f.write(self._model.to_json())
        with open(os.path.join(model_path, "layers_config.txt"), "w") as f:
            f.write(str(len(self._layers_dtbn_intervals)) + '\n')
            for (start, end), d in zip(self._layers_dtbn_intervals, self._distributions):
                f.write(d.__class__.__name__ + '\n' + str(start) + '\n' + str(end) + '\n')
        for i, d in enumerate(self._distributions):
            os.mkdir(os.path.join(model_path, "distribution" + str(i)))
            d.store(os.path.join(model_path, "distribution" + str(i)))
