"""Stein variational gradient descent over M particles (mirrors Pyesian/optimizers/SVGD.py:45-251).
Hyperparameters: lr, M, batch_size; kwarg prior.  All M loss gradients come from one
particle-batched pass; the sequential (Gauss-Seidel) particle sweep, the float64 RBF kernel row,
the repulsion term and the per-particle legacy Adam run on the device.
Multi-GPU: under torch.distributed the particles are sharded by rank and every step all-gathers
the particle matrix over RCCL (Jacobi sweep)."""

import os

import numpy as np

from ..nn.model import model_from_json
from .BBB import ResultTuple
from .Optimizer import DeviceScalar, Optimizer


class Ensemble(list):
    """``SVGD.result()[0]``: the list of particle models, with the ensemble-mean predict."""

    def predict(self, x, *a, **k):
        out = None
        for m in self:
            p = m.predict(x)
            out = p if out is None else out + p
        return out / len(self)


class SVGD(Optimizer):
    def __init__(self):
        super().__init__()
        self._step = 0
        self._M = None
        self._particles = None
        self.train_losses = []
        self.valid_losses = []

    def compile_extra_components(self, **kwargs):
        import torch
        from .. import _lib
        from ..engine import MLPPlan, fill_normal
        self._batch_size = int(self._hyperparameters.batch_size)
        self._prior = kwargs["prior"]
        self._M = int(self._hyperparameters.M)
        self._lr = self._hyperparameters.lr
        # SVGD.py:183: gamma = 1 fixed in the reference.  kernel="median" (or gamma=None) selects the median
        # heuristic of SVGD.baseline__kernel (SVGD.py:165-181, dead code there): bandwidth from the median of the
        # squared distances of the snapshot, hence the Jacobi sweep
        self._gamma = None if kwargs.get("kernel", "rbf") == "median" else kwargs.get("gamma", 1.0)
        if self._gamma is not None:
            self._gamma = float(self._gamma)
        # particle sharding over ranks (one process per GPU); single process = the reference's sweep
        from .. import parallel
        rank, world = parallel.world_info() if kwargs.get("shard", True) else (0, 1)
        self._row0, self._n_local = parallel.shard_range(self._M, world, rank)
        # _force_sharded (tests): take the sharded code path -- gather, separate local rows -- in a world of one rank
        self._sharded = world > 1 or bool(kwargs.get("_force_sharded", False))
        self._sweep = kwargs.get("sweep", "gauss_seidel" if (not self._sharded and self._gamma is not None) else "jacobi")
        if self._sharded and self._sweep != "jacobi":
            raise ValueError("sharded particles need the Jacobi sweep")
        if self._gamma is None and self._sweep != "jacobi":
            raise ValueError("the median-heuristic kernel is evaluated on a snapshot: it needs the Jacobi sweep")
        # overlap_gather: start the all-gather asynchronously on RCCL's stream and wait for it right before its first
        # reader.  Off by default: that ordering has only run with one rank on one GPU (tests/test_gpu_multirank.py),
        # never on a multi-GPU node (no SCALE record yet); PYZ_SVGD_OVERLAP_GATHER=1 or the kwarg opts in.
        self._overlap_gather = bool(kwargs.get("overlap_gather", os.environ.get("PYZ_SVGD_OVERLAP_GATHER", "0") == "1"))
        # overlap_kernel_matrix: the kernel matrix of the snapshot (a function of the particles alone, SVGD.py:183-202) on a
        # second stream while the gradient pass runs (SVGD.py:104-111); only the combine waits for both
        self._overlap_km = bool(kwargs.get("overlap_kernel_matrix", os.environ.get("PYZ_SVGD_OVERLAP_KM", "1") == "1"))
        # shard_gram: the distance pass of the kernel matrix split over the ELEMENTS -- each rank sums the groups of blocks
        # that are its share (D / world of the gathered matrix instead of all of it) and the ranks exchange 8 / world x 32 KB
        # (a second, small all-gather per step); needs a world that divides 8.  Same bits as the local pass (pyz.h).
        # Off by default: on ONE GPU standing in for a rank of eight it measured no faster (the local pass
        # already hides behind the gradient kernels; what is lost there is lost to the cross-stream waits), and it has not
        # run on a multi-GPU node.  PYZ_SVGD_SHARD_GRAM=1 or the kwarg opts in.
        self._shard_gram = (bool(kwargs.get("shard_gram", os.environ.get("PYZ_SVGD_SHARD_GRAM", "0") == "1"))
                            and self._sharded and 8 % world == 0 and self._M % 4 == 0 and self._M <= 64)
        # gather: "rccl" (all_gather_into_tensor) or "p2p" -- every rank writes its rows straight into its peers' matrices
        # (parallel.PeerGather; opt-in: exercised with two processes on one GPU only)
        self._gather = str(kwargs.get("gather", os.environ.get("PYZ_SVGD_GATHER", "rccl"))).lower()
        if self._gather not in ("rccl", "p2p"):
            raise ValueError("gather must be 'rccl' or 'p2p'")
        # every rank draws the same batches and the same particle initialisation: one base seed for all
        self._setup_backend(seed=kwargs.get("seed"), max_particles=self._n_local, chain_per_rank=False)
        self._rank, self._world = rank, world
        self._base_model = self._net
        self._dataset_setup()
        self._num_particles = self._D
        # _init_particles (SVGD.py:143-157): row i = prior samples, drawn here on the device
        mu, rho = self._prior.flat(self._net)
        all_p = torch.empty((self._M, self._D), device="cuda")
        fill_normal(all_p, self._seed, _lib.STREAM_INIT, 0, 0.0, 1.0)
        all_p = all_p * torch.as_tensor(rho).cuda() + torch.as_tensor(mu).cuda()
        self._all = all_p.contiguous()
        self._next = None
        if self._sharded:
            self._local = self._all[self._row0:self._row0 + self._n_local].clone()
        else:
            self._local = self._all                             # same storage
            if self._sweep == "jacobi":                         # one GPU: snapshot and updated matrix alternate (no copy per step)
                self._next = torch.empty_like(self._all)
        self._adam_m = torch.zeros((self._n_local, self._D), device="cuda")
        self._adam_v = torch.zeros((self._n_local, self._D), device="cuda")
        self._loss_dev = torch.zeros(1, device="cuda")
        self._aux = None
        self._peer = None
        if self._sharded and self._gather == "p2p":
            self._peer = parallel.PeerGather(self._M, self._D)
            self._xseq = 0                                      # exchanges so far (its own count: result() exchanges once more)
            self._shard_gram = False
        if self._shard_gram:
            per = 8 // world
            self._g_lo, self._g_hi = rank * per, (rank + 1) * per
            self._groups = torch.zeros((8, 64 * 64), dtype=torch.float64, device="cuda")
            self._groups_local = torch.zeros((per, 64 * 64), dtype=torch.float64, device="cuda") if world > 1 else None
        vx, vy = self._dataset.valid_data.as_numpy()
        self._val_n = len(vx)
        if self._val_n > 0:
            self._vx = torch.as_tensor(np.ascontiguousarray(np.asarray(vx, np.float32).reshape(len(vx), -1))).cuda()
            self._vy = self._labels_to_device(vy)
            self._val_plan = MLPPlan(self._spec, max_batch=self._val_n, max_particles=self._n_local)

    @property
    def _particles_host(self):
        return self._all.cpu().numpy().astype(np.float64)

    def _aux_stream(self):
        import torch
        if self._aux is None:
            self._aux = torch.cuda.Stream()
        return self._aux

    def _kernel_matrix_from_groups(self, snapshot, aux):
        """The kernel matrix with the distance pass sharded over the elements: this rank's groups, the exchange of the group
        sums (all ranks' collectives in program order: the particle gather of this step came first), K rows from all groups."""
        import torch
        from .. import parallel
        with torch.cuda.stream(aux):
            self._plan.svgd_gram_groups(snapshot, self._g_lo, self._g_hi, self._groups, stream=aux)
            if self._groups_local is not None:
                self._groups_local.copy_(self._groups[self._g_lo:self._g_hi])
                parallel.all_gather_rows(self._groups_local, self._groups)
            self._plan.svgd_kernel_matrix_groups(self._groups, snapshot, self._row0, self._n_local, self._gamma, stream=aux)

    def step(self, save_document_path=None):
        import torch
        self._step += 1
        idx, b, _ = self._next_batch()
        from .. import parallel
        # phase 1 -- the loss gradients -- needs the local rows only; the kernel matrix needs the snapshot only; the
        # combine (phi, Adam) needs both.  Several ranks: the all-gather is the one exchange step of the path.
        work = None
        if self._peer is not None:
            # peer-write exchange: this rank's rows go out on the exchange stream; the first reader of the gathered matrix
            # waits for every rank's flag of this step (parked on ITS stream, below)
            self._xseq += 1
            self._all = self._peer.post(self._local, self._row0, self._xseq)
            snapshot, target, cur = self._all, self._local, self._local
        elif self._sharded:
            work = parallel.all_gather_rows(self._local, self._all, async_op=self._overlap_gather,
                                            force_collective=self._world == 1)
            snapshot, target, cur = self._all, self._local, self._local
        elif self._sweep == "jacobi":
            snapshot, target, cur = self._all, self._next, self._all
        else:
            snapshot = target = cur = self._all
        split = (self._sweep == "jacobi" and self._overlap_km
                 and self._plan.svgd_tile_shape(self._n_local, self._M, self._row0))
        if split:
            main, aux = torch.cuda.current_stream(), self._aux_stream()
            if self._peer is not None:
                self._peer.wait(self._xseq, stream=aux)         # the aux stream waits for every rank's rows ...
            elif work is not None:
                with torch.cuda.stream(aux):
                    work.wait()                                 # ... for the gather ...
            else:
                aux.wait_stream(main)                           # ... or for what wrote the snapshot
            if self._shard_gram:
                self._kernel_matrix_from_groups(snapshot, aux)
            else:
                self._plan.svgd_kernel_matrix(snapshot, self._row0, self._n_local, self._gamma, stream=aux)
            done = aux.record_event()
            self._plan.svgd_gradients(cur, self._x_dev, self._y_dev, batch=b, row_idx=idx)
            main.wait_event(done)
            if self._peer is not None:
                main.wait_event(self._peer.copied)              # (the combine rewrites the rows the exchange stream sent)
            if work is not None:
                work.wait()                                     # (the combine reads the snapshot too)
            self._plan.svgd_combine(target, snapshot, self._row0, self._adam_m, self._adam_v, self._lr, self._gamma,
                                    self._step, self._loss_dev)
        else:
            self._plan.svgd_gradients(cur, self._x_dev, self._y_dev, batch=b, row_idx=idx)
            if self._peer is not None:
                self._peer.wait(self._xseq)
                torch.cuda.current_stream().wait_event(self._peer.copied)
            if work is not None:
                work.wait()
            self._plan.svgd_sweep(target, snapshot, self._row0, self._adam_m, self._adam_v, self._lr, self._gamma,
                                  self._step, self._loss_dev, sweep=self._sweep)
        if self._next is not None:                              # one GPU, Jacobi: the updated matrix is the next snapshot
            self._all, self._next = self._next, self._all
            self._local = self._all
        # the step's loss (sum over the LOCAL particles / M) stays on the device.  Several ranks: the sum over ranks is
        # taken on the steps that record it (SVGD.py:137-139) and ONLY there -- a decision every rank takes alike
        # (`verbose` is a per-process argument: a collective that depends on it pairs mismatched calls across ranks);
        # between them the progress bar of a rank shows its own share
        record = self._step % 10 == 0                           # SVGD.py:137-139
        total_loss = self._loss_dev.clone()
        if self._world > 1 and record:
            parallel.sum_over_ranks(total_loss)
        loss = DeviceScalar(total_loss, 0)
        if record:
            # SVGD.py:126-129 forwards the validation split through every particle on every step, but only
            # these steps keep the number: the forward runs when it is observable
            if self._val_n > 0:
                vl, _ = self._val_plan.loss_grad(self._local, self._vx, self._vy, want_grad=False)
                total_val = vl.sum() / self._M
                if self._world > 1:
                    parallel.sum_over_ranks(total_val)
            else:
                total_val = torch.zeros((), device="cuda")
            self.train_losses.append(loss)
            self.valid_losses.append(DeviceScalar(total_val.reshape(1), 0))
        return loss

    def update_parameters_step(self):
        pass

    def result(self):
        import torch
        from .. import parallel
        if self._peer is not None:
            self._xseq += 1                                     # one more exchange: the rows of the last step
            self._all = self._peer.post(self._local, self._row0, self._xseq)
            self._peer.wait(self._xseq)
            torch.cuda.synchronize()
            self._peer.check()
        elif self._sharded:
            parallel.all_gather_rows(self._local, self._all, force_collective=self._world == 1)
        P = self._all.cpu().numpy()
        ensemble = Ensemble()
        for i in range(self._M):                                # SVGD.py:244-249
            m = model_from_json(self._model_config)
            m.set_flat(P[i])
            ensemble.append(m)
        return ResultTuple((ensemble, self.train_losses, self.valid_losses))
