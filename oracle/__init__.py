"""CPU oracle for the Pyesian.optimizers hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement (float64 by default, float32 on request) of
the arithmetic the reference performs in

    Pyesian/optimizers/{SGD,SGLD,HMC,BBB,SVGD}.py  and  Pyesian/nn/BayesianModel.py

written by reading those files as text (line ranges are cited per function).
It exists so that the HIP kernels can be checked against an independent
implementation on the same seeded inputs.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``bayesian_inference_for_nn_amd`` never does and
fails loudly when its HIP library is missing.

PARITY UNPINNED.  The reference's numerics live in TensorFlow 2.15 /
TensorFlow-Probability 0.23, neither of which is installed here (no network),
the reference holds no golden vectors, known-answer tests or asserting tests
for this path, and it cannot be imported.  The
oracle is therefore pinned only by (a) analytic known answers
(tests/test_oracle_kat.py), (b) an independent torch-autograd restatement of
every gradient (oracle/torch_eager.py), and (c) the documented semantics of
the third-party ops listed in SURVEY.md Appendix A.
"""

from . import mlp, philox, sgd, sgld, hmc, bbb, svgd, swag, predict  # noqa: F401
