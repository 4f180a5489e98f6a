"""Hamiltonian Monte Carlo on the full training split (mirrors Pyesian/optimizers/HMC.py:13-187).
Hyperparameters: m, L, epsilon; kwarg prior (GaussianPrior).  Extension: kwarg n_chains runs
that many independent chains in one particle-batched launch (the reference has one)."""

import random

import numpy as np

from ..distributions import Sampled
from ..nn import BayesianModel
from .Optimizer import DeviceScalar, Optimizer


class HMC(Optimizer):
    def __init__(self):
        super().__init__()
        self._nb_burn_epoch = 10
        self._prior = None
        self._frequency = None
        self._samples = None
        self._epsilon = None
        self._L = None
        self._m = None
        self._total_runs = 0
        self._accepted_runs = 0
        self._current_loss = 0

    def compile_extra_components(self, **kwargs):
        import torch
        self._m = self._hyperparameters.m
        self._L = self._hyperparameters.L
        self._epsilon = self._hyperparameters.epsilon
        self._n_chains = int(kwargs.get("n_chains", 1))
        self._merge_ranks = bool(kwargs.get("merge_ranks", True))
        self._setup_backend(seed=kwargs.get("seed"), max_particles=self._n_chains, full_batch=True)
        self._model = self._net
        self._samples = []
        self._frequency = []
        self._prior_spec = kwargs["prior"]
        self._prior = self._prior_spec.get_model_priors(self._net)
        if "nb_burn_epoch" in kwargs:
            # the reference tests one key and reads another (HMC.py:61-62); kept as written
            self._nb_burn_epoch = kwargs["nb_burn_epochs"]
        self._dataset_setup()
        # q <- prior mean (HMC.py:69-72)
        mu, sg = self._prior_spec.flat(self._net)
        if self._prior_spec.is_scalar():
            self._prior_mean, self._prior_sigma = float(self._prior_spec._mean), float(self._prior_spec._std_dev)
            self._pm_vec = self._ps_vec = None
        else:                                                  # per-layer lists: per-element vectors on the device
            self._prior_mean, self._prior_sigma = 0.0, 1.0
            self._pm_vec, self._ps_vec = torch.as_tensor(mu.copy()).cuda(), torch.as_tensor(sg.copy()).cuda()
        self._q = torch.as_tensor(np.repeat(mu[None, :], self._n_chains, axis=0).copy()).cuda()
        self._stats = torch.zeros((self._n_chains, 8), device="cuda")
        # proposals run on a stream of their own: the library replays the launch sequence of a sliced
        # proposal as a hipGraph, and the legacy default stream cannot be captured
        self._stream = torch.cuda.Stream()
        self._step_count = 0
        self._chain_samples = [[] for _ in range(self._n_chains)]
        self._chain_freq = [[] for _ in range(self._n_chains)]
        # proposals are launched back to back; their statistics and state snapshots are read later, in
        # batches (the accept / reject decision itself is taken on the device)
        self._pending = []
        self._defer = False                                   # only train() defers the bookkeeping reads

    def step(self, save_document_path=None, sampling=True, burning=False):
        import torch
        if sampling and any(len(f) == 0 for f in self._chain_freq):
            self._resolve_pending()                             # the start of a chain must see the earlier bookkeeping
            for c in range(self._n_chains):                    # HMC.py:75-77: records the starting q
                if len(self._chain_freq[c]) == 0:
                    self._chain_freq[c].append(1)
                    self._chain_samples[c].append(self._q[c].clone())
        uniforms = [random.random() for _ in range(self._n_chains)]      # HMC.py:91 host Mersenne Twister
        if self._defer:                                       # train() already runs on the side stream
            stats_snap, q_snap = self._launch(uniforms, sampling, burning)
        else:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                stats_snap, q_snap = self._launch(uniforms, sampling, burning)
            torch.cuda.current_stream().wait_stream(self._stream)
        self._step_count += 1
        self._total_runs += 1
        self._pending.append((stats_snap, q_snap, sampling))
        if self._verbose or not self._defer or len(self._pending) >= 64:   # (the progress bar shows every step's accept rate)
            self._resolve_pending()
        return DeviceScalar(stats_snap, 1)

    def _launch(self, uniforms, sampling, burning):
        self._plan.hmc_step(self._q, self._x_dev, self._y_dev, int(self._L), self._epsilon, self._m, self._prior_mean,
                            self._prior_sigma, uniforms, self._step_count, self._seed, self._stats, burning=burning,
                            prior_mean_vec=self._pm_vec, prior_sigma_vec=self._ps_vec)
        return self._stats.clone(), (self._q.clone() if sampling else None)

    def _resolve_pending(self):
        """Acceptance bookkeeping (HMC.py:92-104) of the proposals launched since the last call: one
        device-to-host copy for all of them."""
        import torch
        if not self._pending:
            return
        self._stream.synchronize()
        all_stats = torch.stack([p[0] for p in self._pending]).cpu().numpy()
        if (all_stats[:, :, 7] < 0).any():
            raise RuntimeError("an HMC proposal gave up waiting for its row-slice workgroups (k_hmc_resident: the grid was not "
                               "resident at once); set PYZ_HMC_RESIDENT=0 for one launch per gradient evaluation")
        for (_, q_snap, sampling), stats in zip(self._pending, all_stats):
            accepted = stats[:, 0] != 0
            if accepted[0]:
                self._accepted_runs += 1
            if sampling:
                for c in range(self._n_chains):
                    if accepted[c]:                             # HMC.py:92-96
                        self._chain_freq[c].append(1)
                        self._chain_samples[c].append(q_snap[c])
                    else:                                       # HMC.py:102-103
                        self._chain_freq[c][-1] += 1
            self.last_stats = stats
        self._pending = []
        self._frequency, self._samples = self._chain_freq[0], self._chain_samples[0]

    def train(self, nb_iterations: int, loss_save_document_path: str = None, model_save_frequency: int = None,
              model_save_path: str = None):
        import torch
        self._resolve_pending()
        self._defer = True
        self._stream.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(self._stream):            # (graph replay needs a stream of its own)
                self._train(nb_iterations)
        finally:
            self._defer = False
            self._resolve_pending()
            torch.cuda.current_stream().wait_stream(self._stream)

    def _train(self, nb_iterations: int):
        self._accepted_runs = 0
        self._total_runs = 0
        nb_burn_epoch = self._nb_burn_epoch
        for i in range(nb_burn_epoch):                          # HMC.py:111-116
            loss = self.step(sampling=False, burning=True)
            if self._verbose:                                   # (reading the loss waits for the proposal)
                accept_rate = self._accepted_runs / self._total_runs
                self._print_progress((i + 1) / nb_burn_epoch, suffix="HMC - Burning", loss=loss.numpy().item(),
                                     accept_rate=accept_rate, bar_length=20)
        self._new_progress_line()
        self._resolve_pending()
        self._accepted_runs = 0
        self._total_runs = 0
        self._chain_freq = [[] for _ in range(self._n_chains)]
        self._chain_samples = [[] for _ in range(self._n_chains)]
        for i in range(nb_iterations):                          # HMC.py:121-125
            loss = self.step(sampling=True, burning=False)
            if self._verbose:
                accept_rate = self._accepted_runs / self._total_runs
                self._print_progress((i + 1) / nb_iterations, suffix="HMC - Sampling", loss=loss.numpy().item(),
                                     accept_rate=accept_rate, bar_length=20)
        self._new_progress_line()
        self._resolve_pending()

    def update_parameters_step(self):
        pass

    def result(self) -> BayesianModel:
        """Sampled(samples, frequencies) over all layers (HMC.py:176-187); with n_chains > 1 the chains'
        samples and frequencies are concatenated (independent chains of the same posterior)."""
        self._resolve_pending()
        samples, freqs = [], []
        for c in range(self._n_chains):
            samples += [s.cpu().numpy() for s in self._chain_samples[c]]
            freqs += list(self._chain_freq[c])
        if self._world > 1 and self._merge_ranks:
            # independent chains on every rank: one Sampled posterior over all of them (every rank gets it)
            from .. import parallel
            samples, freqs = parallel.merge_sampled_chains(samples, freqs)
        distribution = Sampled(samples, freqs)
        posterior_model = BayesianModel(self._model_config)
        posterior_model.apply_distribution(distribution, 0, len(self._net.layers) - 1)
        return posterior_model
