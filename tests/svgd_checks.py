"""Checks shared by the SVGD parity tests (not a test module)."""

import numpy as np


def strict_particle_check(p_gpu, st, phis, lr_ts, what):
    """Adam's first steps move an element by ~lr_t * sign(phi) whatever |phi| is, so float32 and float64 may
    differ by up to 2 lr_t per step on elements whose phi is (numerically) zero -- and ONLY there: elements
    with |phi_oracle| > 1e-6 max|phi| in every step must agree to 2e-4 of the particle scale."""
    err = np.abs(p_gpu.cpu().numpy().astype(np.float64) - st.particles)
    strong = np.ones(err.shape, dtype=bool)
    for phi in phis:
        strong &= np.abs(phi) > 1e-6 * np.abs(phi).max()
    scale = np.abs(st.particles).max()
    assert strong.mean() > 0.5, (what, strong.mean())
    assert err[strong].max() <= 2e-4 * scale, f"{what}: {err[strong].max():.3e} on elements with a definite phi (scale {scale:.3e})"
    if (~strong).any():
        assert err[~strong].max() <= 2.0 * sum(lr_ts) * 1.001 + 2e-4 * scale, f"{what}: {err[~strong].max():.3e} on phi ~ 0 elements"


def lr_t(lr, t):
    return lr * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
