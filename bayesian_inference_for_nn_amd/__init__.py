"""MI355X (gfx950) backend for the Pyesian.optimizers hot path.

Hand-written HIP kernels behind the C-ABI of include/pyz.h (csrc/libpyz.so),
driven from a Python surface that mirrors the reference's
``Optimizer.compile()/train()/result()`` + ``BayesianModel`` API.
There is no CPU fallback: without the HIP library and an MI355X every
numerical entry point raises.
"""

__version__ = "0.1.0"
