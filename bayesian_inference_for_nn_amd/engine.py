"""Host-side handle over the C-ABI plan (`pyz_mlp`).  torch is used for device
memory and streams only; every numerical operation is a HIP kernel of libpyz.so.
All shapes are validated here before a pointer is handed to a kernel."""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ACT, LOSS, SWEEP, check, ptr

SVGD_GROUPS = 8   # PYZ_SVGD_GROUPS of include/pyz.h


@dataclass(frozen=True)
class MLPSpec:
    """dims = [in, h1, ..., out]; acts[l] = Keras activation name of Dense layer l."""
    dims: Tuple[int, ...]
    acts: Tuple[str, ...]
    loss: str = "scce"

    @property
    def n_layers(self):
        return len(self.acts)

    @property
    def n_params(self):
        return sum((i + 1) * o for i, o in zip(self.dims[:-1], self.dims[1:]))

    def layer_slices(self):
        out, off = [], 0
        for i, o in zip(self.dims[:-1], self.dims[1:]):
            out.append(slice(off, off + (i + 1) * o))
            off += (i + 1) * o
        return out

    def variable_shapes(self):
        """[(kernel shape, bias shape)] per layer, in flat order."""
        return [((i, o), (o,)) for i, o in zip(self.dims[:-1], self.dims[1:])]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gamma(gamma) -> float:
    """Fixed RBF bandwidth, or None / "median" for the median heuristic (PYZ_SVGD_GAMMA_MEDIAN)."""
    if gamma is None or gamma == "median":
        return _lib.GAMMA_MEDIAN
    if not float(gamma) > 0.0:
        raise ValueError("gamma must be positive, or None / 'median' for the median heuristic")
    return float(gamma)


def _f32(t: torch.Tensor, shape=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA(HIP) torch tensor")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f"{name} must be contiguous float32")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


class MLPPlan:
    """Owns one `pyz_mlp` handle (device workspace for up to max_batch rows x max_particles)."""

    def __init__(self, spec: MLPSpec, max_batch: int, max_particles: int = 1, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("bayesian_inference_for_nn_amd needs an MI355X (no HIP device visible); there is no CPU path")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.spec, self.max_batch, self.max_particles = spec, int(max_batch), int(max_particles)
        dims = (C.c_int32 * len(spec.dims))(*spec.dims)
        acts = (C.c_int32 * len(spec.acts))(*[ACT[a] for a in spec.acts])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.pyz_mlp_create(spec.n_layers, dims, acts, LOSS[spec.loss], self.max_batch,
                                          self.max_particles, C.byref(h)))
        self.h = h
        self.D = int(self.lib.pyz_mlp_param_count(h))
        assert self.D == spec.n_params

    def close(self):
        if getattr(self, "h", None):
            self.lib.pyz_mlp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check_xy(self, x, y, row_idx, batch):
        _f32(x, name="x")
        if x.dim() != 2 or x.shape[1] != self.spec.dims[0]:
            raise ValueError(f"x must be (rows, {self.spec.dims[0]})")
        n_rows = x.shape[0]
        if row_idx is not None:
            if row_idx.dtype != torch.int32 or not row_idx.is_cuda or not row_idx.is_contiguous():
                raise TypeError("row_idx must be a contiguous CUDA int32 tensor")
        elif batch > n_rows:
            raise ValueError("batch exceeds the rows of x")
        if y is not None:
            if self.spec.loss == "scce":
                if y.dtype != torch.int32 or not y.is_cuda or y.numel() != n_rows:
                    raise TypeError("labels must be CUDA int32, one per row of x")
            else:
                _f32(y, name="y")
                if y.numel() != n_rows * self.spec.dims[-1]:
                    raise ValueError("targets must be (rows, out)")
        if not (1 <= batch <= self.max_batch):
            raise ValueError(f"batch {batch} outside [1, {self.max_batch}]")

    # ------------------------------------------------------------------ G1-G3
    def forward(self, theta, x, batch=None, row_idx=None):
        P = 1 if theta.dim() == 1 else theta.shape[0]
        _f32(theta, name="theta")
        assert theta.numel() == P * self.D
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, None, row_idx, batch)
        if row_idx is not None:
            assert row_idx.numel() >= batch
        out = torch.empty((P, batch, self.spec.dims[-1]), dtype=torch.float32, device=self.device)
        check(self.lib.pyz_mlp_forward(self.h, ptr(theta), P, ptr(x), ptr(row_idx), batch, ptr(out), _stream()))
        return out

    def loss_grad(self, theta, x, y, batch=None, row_idx=None, want_grad=True):
        P = 1 if theta.dim() == 1 else theta.shape[0]
        _f32(theta, name="theta")
        assert theta.numel() == P * self.D
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        if row_idx is not None:
            assert row_idx.numel() >= batch
        grad = torch.empty((P, self.D), dtype=torch.float32, device=self.device) if want_grad else None
        loss = torch.empty((P,), dtype=torch.float32, device=self.device)
        check(self.lib.pyz_mlp_loss_grad(self.h, ptr(theta), P, ptr(x), ptr(y), ptr(row_idx), batch, ptr(grad),
                                         ptr(loss), _stream()))
        return loss, grad

    # ------------------------------------------------------------------ S1
    def sgd_step(self, theta, x, y, lr, loss_out, batch=None, row_idx=None):
        _f32(theta, (self.D,), "theta")
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        check(self.lib.pyz_sgd_step(self.h, ptr(theta), ptr(x), ptr(y), ptr(row_idx), batch, float(lr), ptr(loss_out),
                                    _stream()))

    def swag_step(self, theta, mean, sq_mean, dev_row, x, y, lr, n, update_moments, loss_out, batch=None, row_idx=None):
        for t, nm in ((theta, "theta"), (mean, "mean"), (sq_mean, "sq_mean")):
            _f32(t, (self.D,), nm)
        if dev_row is not None:
            _f32(dev_row, (self.D,), "dev_row")
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        check(self.lib.pyz_swag_step(self.h, ptr(theta), ptr(mean), ptr(sq_mean), ptr(dev_row), ptr(x), ptr(y),
                                     ptr(row_idx), batch, float(lr), int(n), 1 if update_moments else 0, ptr(loss_out),
                                     _stream()))

    # ------------------------------------------------------------------ L2/L3
    def sgld_step(self, theta, mean, sq_mean, x, y, lr, n, seed, loss_out, batch=None, row_idx=None, unit_noise=None):
        for t, nm in ((theta, "theta"), (mean, "mean"), (sq_mean, "sq_mean")):
            _f32(t, (self.D,), nm)
        if unit_noise is not None:
            _f32(unit_noise, (self.D,), "unit_noise")
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        check(self.lib.pyz_sgld_step(self.h, ptr(theta), ptr(mean), ptr(sq_mean), ptr(x), ptr(y), ptr(row_idx), batch,
                                     float(lr), int(n), int(seed), ptr(unit_noise), ptr(loss_out), _stream()))

    def sgld_run(self, theta, mean, sq_mean, x, y, row_idx, batch_sizes: Sequence[int], lrs: Sequence[float], n0, seed,
                 losses_out, use_graph=True, slot0=0):
        """row_idx: int32 (slots, max_batch) on the device; step s of this call uses slot slot0+s of
        row_idx / losses_out; batch_sizes / lrs: host sequences for the n_steps of this call."""
        for t, nm in ((theta, "theta"), (mean, "mean"), (sq_mean, "sq_mean")):
            _f32(t, (self.D,), nm)
        n_steps = len(batch_sizes)
        assert len(lrs) == n_steps and n_steps > 0
        self._check_xy(x, y, row_idx, 1)
        if slot0 < 0 or row_idx.numel() < (slot0 + n_steps) * self.max_batch:
            raise ValueError("row_idx must hold (slot0 + n_steps) * max_batch entries")
        _f32(losses_out, name="losses_out")
        assert losses_out.numel() >= slot0 + n_steps
        if any(int(b) < 1 or int(b) > self.max_batch for b in batch_sizes):
            raise ValueError("batch size outside the plan")
        bs = (C.c_int32 * n_steps)(*[int(b) for b in batch_sizes])
        lr = (C.c_float * n_steps)(*[float(v) for v in lrs])
        check(self.lib.pyz_sgld_run(self.h, ptr(theta), ptr(mean), ptr(sq_mean), ptr(x), ptr(y), ptr(row_idx), bs, lr,
                                    n_steps, int(n0), int(slot0), int(seed), ptr(losses_out), 1 if use_graph else 0,
                                    _stream()))

    def sgd_run(self, theta, x, y, row_idx, batch_sizes, lrs, losses_out, use_graph=True, slot0=0):
        """n = len(batch_sizes) SGD steps without host work in between (see sgld_run for the table layout)."""
        n_steps = len(batch_sizes)
        assert len(lrs) == n_steps and n_steps > 0
        _f32(theta, (self.D,), "theta")
        self._check_xy(x, y, row_idx, 1)
        if slot0 < 0 or row_idx.numel() < (slot0 + n_steps) * self.max_batch or losses_out.numel() < slot0 + n_steps:
            raise ValueError("row_idx / losses_out too small")
        if any(int(b) < 1 or int(b) > self.max_batch for b in batch_sizes):
            raise ValueError("batch size outside the plan")
        bs = (C.c_int32 * n_steps)(*[int(b) for b in batch_sizes])
        lr = (C.c_float * n_steps)(*[float(v) for v in lrs])
        check(self.lib.pyz_sgd_run(self.h, ptr(theta), ptr(x), ptr(y), ptr(row_idx), bs, lr, n_steps, int(slot0),
                                   ptr(losses_out), 1 if use_graph else 0, _stream()))

    def swag_run(self, theta, mean, sq_mean, dev, frequency, x, y, row_idx, batch_sizes, lrs, n0, losses_out,
                 use_graph=True, slot0=0):
        """n = len(batch_sizes) SWAG steps without host work in between; dev is the (k, D) deviation matrix."""
        n_steps = len(batch_sizes)
        assert len(lrs) == n_steps and n_steps > 0
        for t, nm in ((theta, "theta"), (mean, "mean"), (sq_mean, "sq_mean")):
            _f32(t, (self.D,), nm)
        _f32(dev, name="dev")
        assert dev.dim() == 2 and dev.shape[1] == self.D
        self._check_xy(x, y, row_idx, 1)
        if slot0 < 0 or row_idx.numel() < (slot0 + n_steps) * self.max_batch or losses_out.numel() < slot0 + n_steps:
            raise ValueError("row_idx / losses_out too small")
        if any(int(b) < 1 or int(b) > self.max_batch for b in batch_sizes):
            raise ValueError("batch size outside the plan")
        bs = (C.c_int32 * n_steps)(*[int(b) for b in batch_sizes])
        lr = (C.c_float * n_steps)(*[float(v) for v in lrs])
        check(self.lib.pyz_swag_run(self.h, ptr(theta), ptr(mean), ptr(sq_mean), ptr(dev), int(dev.shape[0]), int(frequency),
                                    ptr(x), ptr(y), ptr(row_idx), bs, lr, n_steps, int(n0), int(slot0), ptr(losses_out),
                                    1 if use_graph else 0, _stream()))

    def last_run_path(self):
        """("graph" | "eager" | "mixed", steps) of the last sgld_run / sgd_run / swag_run call on this plan."""
        g, e, n = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.pyz_last_run_info(self.h, C.byref(g), C.byref(e), C.byref(n)))
        kind = "graph" if e.value == 0 else ("eager" if g.value == 0 else "mixed")
        return kind, g.value + e.value

    def last_run_graph_launches(self) -> int:
        n = C.c_int32()
        check(self.lib.pyz_last_run_info(self.h, None, None, C.byref(n)))
        return n.value

    def check_finite(self):
        """Synchronises the current stream; raises PyzError (code E_NAN) if a step since the last check produced a
        NaN / Inf loss."""
        check(self.lib.pyz_check_finite(self.h, _stream()))

    # ------------------------------------------------------------------ B2-B4
    def bbb_step(self, mu, rho, w, x, y, lr, alpha, prior_mean, prior_rho, step, seed, cost_out, batch=None,
                 row_idx=None, eps=None, prior_mean_vec=None, prior_rho_vec=None):
        for t, nm in ((mu, "mu"), (rho, "rho"), (w, "w")):
            _f32(t, (self.D,), nm)
        for t, nm in ((eps, "eps"), (prior_mean_vec, "prior_mean_vec"), (prior_rho_vec, "prior_rho_vec")):
            if t is not None:
                _f32(t, (self.D,), nm)
        _f32(cost_out, name="cost_out")
        assert cost_out.numel() >= 3
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        check(self.lib.pyz_bbb_step(self.h, ptr(mu), ptr(rho), ptr(w), ptr(x), ptr(y), ptr(row_idx), batch, float(lr),
                                    float(alpha), float(prior_mean), float(prior_rho), ptr(prior_mean_vec),
                                    ptr(prior_rho_vec), int(step), int(seed), ptr(eps), ptr(cost_out), _stream()))

    def bbb_run(self, mu, rho, w, x, y, row_idx, batch_sizes, lrs, alpha, prior_mean, prior_rho, step0, seed, costs_out,
                use_graph=True, slot0=0, prior_mean_vec=None, prior_rho_vec=None, val_plan=None, val_x=None, val_y=None,
                val_losses_out=None):
        """n = len(batch_sizes) BBB steps without host work in between (see sgld_run for the table layout); costs_out
        (slots, 4): {cost, data loss, log q - log p, -} per step.  val_plan / val_x / val_y / val_losses_out (slots):
        the validation forward of BBB.py:203-209 inside the run (steps with step % 10 != 0)."""
        n_steps = len(batch_sizes)
        assert len(lrs) == n_steps and n_steps > 0
        for t, nm in ((mu, "mu"), (rho, "rho"), (w, "w")):
            _f32(t, (self.D,), nm)
        for t, nm in ((prior_mean_vec, "prior_mean_vec"), (prior_rho_vec, "prior_rho_vec")):
            if t is not None:
                _f32(t, (self.D,), nm)
        self._check_xy(x, y, row_idx, 1)
        _f32(costs_out, name="costs_out")
        if slot0 < 0 or row_idx.numel() < (slot0 + n_steps) * self.max_batch or costs_out.numel() < 4 * (slot0 + n_steps):
            raise ValueError("row_idx / costs_out too small")
        if any(int(b) < 1 or int(b) > self.max_batch for b in batch_sizes):
            raise ValueError("batch size outside the plan")
        n_val = 0
        if val_plan is not None:
            _f32(val_x, name="val_x")
            n_val = int(val_x.shape[0])
            if val_x.shape[1] != self.spec.dims[0] or n_val > val_plan.max_batch or val_plan.D != self.D:
                raise ValueError("validation split does not fit the validation plan")
            _f32(val_losses_out, name="val_losses_out")
            if val_losses_out.numel() < slot0 + n_steps:
                raise ValueError("val_losses_out too small")
        bs = (C.c_int32 * n_steps)(*[int(b) for b in batch_sizes])
        lr = (C.c_float * n_steps)(*[float(v) for v in lrs])
        check(self.lib.pyz_bbb_run(self.h, ptr(mu), ptr(rho), ptr(w), ptr(x), ptr(y), ptr(row_idx), bs, lr, n_steps, float(alpha),
                                   float(prior_mean), float(prior_rho), ptr(prior_mean_vec), ptr(prior_rho_vec), int(step0),
                                   int(slot0), int(seed), ptr(costs_out), val_plan.h if val_plan is not None else None,
                                   ptr(val_x), ptr(val_y), n_val, ptr(val_losses_out), 1 if use_graph else 0, _stream()))

    # ------------------------------------------------------------------ H2-H5
    def hmc_step(self, q, x, y, L, epsilon, m, prior_mean, prior_sigma, uniforms, step, seed, stats_out, burning=False,
                 unit_p=None, prior_mean_vec=None, prior_sigma_vec=None):
        P = 1 if q.dim() == 1 else q.shape[0]
        _f32(q, name="q")
        assert q.numel() == P * self.D
        if unit_p is not None:
            _f32(unit_p, name="unit_p")
            assert unit_p.numel() == P * self.D
        for t, nm in ((prior_mean_vec, "prior_mean_vec"), (prior_sigma_vec, "prior_sigma_vec")):
            if t is not None:
                _f32(t, (self.D,), nm)
        _f32(stats_out, name="stats_out")
        assert stats_out.numel() >= 8 * P
        n_rows = x.shape[0]
        self._check_xy(x, y, None, n_rows)
        u = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float32).reshape(-1))
        assert u.size == P
        check(self.lib.pyz_hmc_step(self.h, ptr(q), P, ptr(x), ptr(y), n_rows, int(L), float(epsilon), float(m),
                                    float(prior_mean), float(prior_sigma), ptr(prior_mean_vec), ptr(prior_sigma_vec),
                                    1 if burning else 0, u.ctypes.data_as(C.POINTER(C.c_float)), int(step), int(seed),
                                    ptr(unit_p), ptr(stats_out), _stream()))

    # ------------------------------------------------------------------ V2-V4
    def svgd_step(self, particles, all_particles, row0, adam_m, adam_v, x, y, lr, gamma, t, loss_out, sweep="gauss_seidel",
                  batch=None, row_idx=None):
        _f32(particles, name="particles")
        _f32(all_particles, name="all_particles")
        n_local, n_total = particles.shape[0], all_particles.shape[0]
        assert particles.shape[1] == self.D and all_particles.shape[1] == self.D
        _f32(adam_m, (n_local, self.D), "adam_m")
        _f32(adam_v, (n_local, self.D), "adam_v")
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        check(self.lib.pyz_svgd_step(self.h, ptr(particles), n_local, ptr(all_particles), n_total, int(row0),
                                     ptr(adam_m), ptr(adam_v), ptr(x), ptr(y), ptr(row_idx), batch, float(lr),
                                     _gamma(gamma), int(t), SWEEP[sweep], ptr(loss_out), _stream()))

    def svgd_gradients(self, particles, x, y, batch=None, row_idx=None):
        """Phase 1 of svgd_step: the loss gradients of the local particles (kept inside the plan)."""
        _f32(particles, name="particles")
        assert particles.dim() == 2 and particles.shape[1] == self.D
        batch = int(batch if batch is not None else (row_idx.numel() if row_idx is not None else x.shape[0]))
        self._check_xy(x, y, row_idx, batch)
        check(self.lib.pyz_svgd_gradients(self.h, ptr(particles), particles.shape[0], ptr(x), ptr(y), ptr(row_idx), batch,
                                          _stream()))

    def svgd_sweep(self, particles, all_particles, row0, adam_m, adam_v, lr, gamma, t, loss_out, sweep="gauss_seidel"):
        """Phase 2: kernel rows, repulsion and Adam on the gradients svgd_gradients left; first reader of all_particles."""
        _f32(particles, name="particles")
        _f32(all_particles, name="all_particles")
        n_local, n_total = particles.shape[0], all_particles.shape[0]
        assert particles.shape[1] == self.D and all_particles.shape[1] == self.D
        _f32(adam_m, (n_local, self.D), "adam_m")
        _f32(adam_v, (n_local, self.D), "adam_v")
        check(self.lib.pyz_svgd_sweep(self.h, ptr(particles), n_local, ptr(all_particles), n_total, int(row0), ptr(adam_m),
                                      ptr(adam_v), float(lr), _gamma(gamma), int(t), SWEEP[sweep], ptr(loss_out), _stream()))

    def svgd_tile_shape(self, n_local, n_total, row0) -> bool:
        """Shapes pyz_svgd_kernel_matrix / pyz_svgd_combine take (the all-rows-at-once Jacobi kernels)."""
        return n_total <= 64 and n_local % 4 == 0 and row0 % 4 == 0

    def svgd_kernel_matrix(self, all_particles, row0, n_local, gamma, stream=None):
        """First half of the Jacobi sweep: K rows of the local particles against the snapshot (kept inside the plan).
        Touches neither gradients nor losses: may run on `stream` while svgd_gradients runs on another."""
        _f32(all_particles, name="all_particles")
        assert all_particles.dim() == 2 and all_particles.shape[1] == self.D
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
        check(self.lib.pyz_svgd_kernel_matrix(self.h, ptr(all_particles), all_particles.shape[0], int(row0), int(n_local),
                                              _gamma(gamma), st))

    def svgd_gram_groups(self, all_particles, g_lo, g_hi, groups, stream=None):
        """The distance pass over groups [g_lo, g_hi) of the SVGD_GROUPS groups of element blocks, all pairs: the group sums
        go to rows [g_lo, g_hi) of `groups` (SVGD_GROUPS, 64 * 64) float64 (a rank's share of a D-sharded pass)."""
        _f32(all_particles, name="all_particles")
        assert all_particles.dim() == 2 and all_particles.shape[1] == self.D
        assert groups.dtype == torch.float64 and groups.is_cuda and groups.is_contiguous() and groups.numel() == SVGD_GROUPS * 4096
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
        check(self.lib.pyz_svgd_gram_groups(self.h, ptr(all_particles), all_particles.shape[0], int(g_lo), int(g_hi),
                                            ptr(groups), st))

    def svgd_kernel_matrix_groups(self, groups, all_particles, row0, n_local, gamma, stream=None):
        """svgd_kernel_matrix from the complete group sums (every rank's svgd_gram_groups, gathered) instead of a pass over
        the snapshot; same bits."""
        _f32(all_particles, name="all_particles")
        assert groups.dtype == torch.float64 and groups.is_cuda and groups.is_contiguous() and groups.numel() == SVGD_GROUPS * 4096
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
        check(self.lib.pyz_svgd_kernel_matrix_groups(self.h, ptr(groups), ptr(all_particles), all_particles.shape[0], int(row0),
                                                     int(n_local), _gamma(gamma), st))

    def svgd_combine(self, particles, all_particles, row0, adam_m, adam_v, lr, gamma, t, loss_out):
        """Second half: phi, Adam and the loss of every local row; after svgd_gradients AND svgd_kernel_matrix.
        `particles` (n_local, D) is only written (the rows' current values are the snapshot's)."""
        _f32(particles, name="particles")
        _f32(all_particles, name="all_particles")
        n_local, n_total = particles.shape[0], all_particles.shape[0]
        assert particles.shape[1] == self.D and all_particles.shape[1] == self.D
        _f32(adam_m, (n_local, self.D), "adam_m")
        _f32(adam_v, (n_local, self.D), "adam_v")
        check(self.lib.pyz_svgd_combine(self.h, ptr(particles), n_local, ptr(all_particles), n_total, int(row0), ptr(adam_m),
                                        ptr(adam_v), float(lr), _gamma(gamma), int(t), ptr(loss_out), _stream()))

    # ------------------------------------------------------------------ R1
    def predict(self, weights, x, want_samples=True):
        _f32(weights, name="weights")
        S = weights.shape[0]
        assert weights.dim() == 2 and weights.shape[1] == self.D
        _f32(x, name="x")
        n = x.shape[0]
        assert x.shape[1] == self.spec.dims[0] and n <= self.max_batch
        C_out = self.spec.dims[-1]
        samples = torch.empty((S, n, C_out), dtype=torch.float32, device=self.device) if want_samples else None
        mean = torch.empty((n, C_out), dtype=torch.float32, device=self.device)
        check(self.lib.pyz_predict(self.h, ptr(weights), S, ptr(x), n, ptr(samples), ptr(mean), _stream()))
        return samples, mean


class KernelProbe:
    """with KernelProbe(max_launches) as kp: ...launch steps...  -> kp.launches = [(kernel expression, microseconds)]
    in launch order: each kernel's own begin / end timestamps (measurement only; runs launch eagerly inside)."""

    def __init__(self, max_launches: int = 4096):
        self.max = int(max_launches)
        self.launches = []

    def __enter__(self):
        check(_lib.load().pyz_probe_begin(self.max))
        return self

    def __exit__(self, *exc):
        us = (C.c_float * self.max)()
        names = C.create_string_buffer(self.max * 64)
        n = C.c_int()
        rc = _lib.load().pyz_probe_end(_stream(), us, names, 64, C.byref(n))
        if exc[0] is None:
            check(rc)
            raw = names.raw
            self.launches = [(raw[64 * i:64 * i + 64].split(b"\0", 1)[0].decode(), float(us[i])) for i in range(n.value)]
        return False

    def by_kernel(self):
        """{kernel expression: (launch count, mean microseconds)}"""
        acc = {}
        for name, v in self.launches:
            c, t = acc.get(name, (0, 0.0))
            acc[name] = (c + 1, t + v)
        return {k: (c, t / c) for k, (c, t) in acc.items()}


def fill_normal(out: torch.Tensor, seed: int, stream_id: int, step: int, mean: float = 0.0, std: float = 1.0):
    _f32(out, name="out")
    lib = _lib.load()
    check(lib.pyz_fill_normal(ptr(out), out.numel(), int(seed), int(stream_id), int(step), float(mean), float(std),
                              _stream()))
    return out


def sample_normal_rows(out: torch.Tensor, col0: int, loc: torch.Tensor, scale: torch.Tensor, seed: int, first_draw: int):
    """out[r, col0:col0+len] = loc + scale * z_r for every row r of the (n, stride) CUDA matrix `out`."""
    lib = _lib.load()
    _f32(out, name="out")
    _f32(loc, name="loc")
    _f32(scale, (loc.numel(),), "scale")
    assert out.dim() == 2
    check(lib.pyz_sample_normal_rows(ptr(out), out.shape[0], out.shape[1], int(col0), loc.numel(), ptr(loc), ptr(scale),
                                     int(seed), _lib.STREAM_PREDICT, int(first_draw) & 0xFFFFFFFF, _stream()))
