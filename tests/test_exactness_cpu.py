"""CPU-side checks of what the blend kernels' exact threshold decisions rest on (csrc/gs_common.h, "threshold decisions"):

* the algorithm of ``gs_exp_cr`` -- range reduction, degree-13 Taylor polynomial, ldexp, all in double -- agrees with glibc's
  ``(float)exp((double)x)`` (the oracle's and the emulated reference run's definition) on a dense sweep of fp32 inputs;
* the proven distance between the kernels' alpha and the reference's, ``|ln a_kernel - ln a_reference| <= u (2 |e| + 9)``, holds on
  random conics (needles included), pixels and opacities in a NumPy model of both evaluations: the reference's fp32 operation
  order (UTL:281-283 forward, UTL:336-339 backward) with the correctly rounded exponential on one side, the same exponent,
  ``2^(e * log2 e)`` with a one-ulp hardware exponential and ``amp = fl(opacity * rescale)`` on the other;
* the exponents of the two sides are bit-identical (what makes the bound independent of the conic's condition).
"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
U = 2.0 ** -24


def test_exp_cr_algorithm_matches_glibc_on_a_dense_sweep(tmp_path):
    exe = str(tmp_path / "exp_cr_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", os.path.join(HERE, "native", "exp_cr_check.c"), "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe, "389"], check=True, capture_output=True, text=True).stdout
    n, bad = (int(tok.split("=")[1]) for tok in out.split())
    assert n > 5_000_000 and bad == 0, out


def _scene(n, rng):
    """Random screen-space Gaussians (anisotropy up to 100, any orientation, the +0.3 low-pass of UTL:263-264) and one pixel
    each, placed so that alpha spreads over [1e-3, 1]."""
    s1 = np.exp(rng.uniform(np.log(0.3), np.log(60), n))
    s2 = s1 * np.exp(rng.uniform(np.log(0.01), 0, n))
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    c00 = c * c * s1 * s1 + s * s * s2 * s2 + 0.3
    c11 = s * s * s1 * s1 + c * c * s2 * s2 + 0.3
    c01 = c * s * (s1 * s1 - s2 * s2)
    det = c00 * c11 - c01 * c01
    A, B, C = (c11 / det).astype(f32), (-c01 / det).astype(f32), (c00 / det).astype(f32)
    opacity, rescale = rng.uniform(0.02, 1.0, n).astype(f32), rng.uniform(0.3, 1.0, n).astype(f32)
    target = rng.uniform(0, 7.0, n)
    d = rng.uniform(0, 2 * np.pi, n)
    ddx, ddy = np.cos(d), np.sin(d)
    q = 0.5 * (A * ddx * ddx + C * ddy * ddy) + B * ddx * ddy
    r = np.sqrt(target / np.maximum(q, 1e-30))
    u0, v0 = rng.uniform(0, 1900, n).astype(f32), rng.uniform(0, 1000, n).astype(f32)
    px, py = (np.floor(u0 + r * ddx) + 0.5).astype(f32), (np.floor(v0 + r * ddy) + 0.5).astype(f32)
    return (px - u0).astype(f32), (py - v0).astype(f32), A, B, C, opacity, rescale


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def test_alpha_bound_between_kernel_and_reference():
    rng = np.random.default_rng(0)
    n = 2_000_000
    dx, dy, A, B, C, opacity, rescale = _scene(n, rng)
    # reference, forward pass (UTL:281-284, RAS:447): every fp32 operation rounded, exp correctly rounded
    e_ref = (f32(-0.5) * ((dx * dx) * A + (dy * dy) * C) - (dx * dy) * B).astype(f32)
    a_ref = ((np.exp(e_ref.astype(np.float64)).astype(f32) * rescale) * opacity).astype(f32)
    # reference, backward pass (UTL:336-341)
    m0, m1 = (A * dx + B * dy).astype(f32), (B * dx + C * dy).astype(f32)
    e_bwd = (f32(-0.5) * (dx * m0 + dy * m1)).astype(f32)
    a_bwd = ((np.exp(e_bwd.astype(np.float64)).astype(f32) * rescale) * opacity).astype(f32)
    # kernels: the same exponents (gs_pair_alpha_forward ends in one fma: -0.5 * t is exact, so it rounds as the subtraction does)
    t = ((dx * dx) * A + (dy * dy) * C).astype(f32)
    e_k = _fma(np.full(n, f32(-0.5)), t, -((dx * dy) * B).astype(f32))
    assert np.array_equal(e_k, e_ref), "the forward exponent must be the reference's to the last bit"
    s = (dx * m0 + dy * m1).astype(f32)
    log2e = f32(1.4426950408889634)
    assert np.array_equal((s * (f32(-0.5) * log2e)).astype(f32), (e_bwd * log2e).astype(f32)), \
        "s * (-0.5 log2 e) must equal (-0.5 s) * log2 e bit for bit (gs_pair_alpha_backward)"
    amp = (opacity * rescale).astype(f32)
    for name, e, a_reference in (("forward", e_ref, a_ref), ("backward", e_bwd, a_bwd)):
        hw = np.exp2((e * log2e).astype(f32).astype(np.float64)).astype(f32)
        # a one-ulp hardware exponential: the correctly rounded value moved one ulp either way at random
        hw = np.nextafter(hw, np.where(rng.random(n) < 0.5, f32(0), f32(np.inf)).astype(f32)).astype(f32)
        a_kernel = (hw * amp).astype(f32)
        ok = (a_reference > 1e-3) & (a_kernel > 0)
        d = np.abs(np.log(a_kernel[ok].astype(np.float64)) - np.log(a_reference[ok].astype(np.float64)))
        bound = U * (2.0 * np.abs(e[ok].astype(np.float64)) + 9.0)
        worst = float((d / bound).max())
        print(f"[parity] alpha_bound.{name}: largest distance / proven bound = {worst:.3f} over {int(ok.sum())} pairs")
        assert worst <= 1.0, (name, worst)
