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


def test_the_exponent_bound_of_the_hit_test_is_on_the_safe_side():
    """Round 6 (csrc/gs_common.h, "the 1/255 decision in the exponent's domain"): the blend kernels skip a (pixel, Gaussian) pair
    whose exponent is below ``e_lo = L~ - h``, ``L~ = -logf(255 amp)``, ``h = 4/3 u (6 + 4 |L~|)``, BEFORE evaluating any
    exponential.  That is only right if the reference never blends such a pair.  For random (opacity, rescale) the exact
    threshold e* of the reference's decision -- the smallest fp32 exponent with fl(fl(exp_cr(e) rescale) opacity) >= fl(1/255),
    found by bisection over the fp32 numbers, the reference's alpha being monotone in e -- must lie at or above e_lo, and not
    far above it (the uncertain zone [e_lo, e*) is what costs exact re-evaluations: a few 1e-6 wide).  NumPy's float32 log
    stands in for the device's logf (both within an ulp or two of the true value: the bound charges two)."""
    rng = np.random.default_rng(3)
    n = 200_000
    opacity = rng.uniform(1e-3, 1.0, n).astype(f32)
    rescale = np.concatenate([rng.uniform(0.05, 1.0, n // 2), np.exp(rng.uniform(np.log(1e-6), 0.0, n - n // 2))]).astype(f32)
    amp = (opacity * rescale).astype(f32)
    L = (-np.log((f32(255.0) * amp).astype(f32))).astype(f32)
    e_lo = (L - f32(4.0 / 3.0) * f32(U) * (f32(6.0) + f32(4.0) * np.abs(L))).astype(f32)
    eps = f32(1.0 / 255.0)

    def reference_hits(e):   # UTL:284 then RAS:447, exp correctly rounded (double exp rounded once)
        return ((np.exp(e.astype(np.float64)).astype(f32) * rescale).astype(f32) * opacity).astype(f32) >= eps

    # bisection on the ordered integers behind the fp32 values of [-30, 30]
    def to_ord(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, np.int64(-(2 ** 31)) - i - 1 + 0, i)

    def from_ord(o):
        i = np.where(o < 0, np.int64(-(2 ** 31)) - o - 1, o).astype(np.int32)
        return i.view(f32)
    lo, hi = to_ord(np.full(n, -30.0, f32)), to_ord(np.full(n, 30.0, f32))
    assert not reference_hits(from_ord(lo)).any() and reference_hits(from_ord(hi)).all()
    while (hi - lo > 1).any():
        mid = (lo + hi) // 2
        hit = reference_hits(from_ord(mid))
        hi = np.where(hit, mid, hi)
        lo = np.where(hit, lo, mid)
    e_star = from_ord(hi)                        # the smallest exponent the reference blends
    assert reference_hits(e_star).all() and not reference_hits(from_ord(hi - 1)).any()
    assert (e_lo <= e_star).all(), float((e_lo - e_star).max())
    zone = (e_star.astype(np.float64) - e_lo.astype(np.float64))
    print(f"[parity] hit_exponent_bound: uncertain zone below the reference's threshold: max {zone.max():.2e}, mean {zone.mean():.2e} "
          f"(|L| up to {float(np.abs(L).max()):.1f})")
    assert zone.max() < 4e-5 and zone.mean() < 1e-5
    # and the margin actually left (in units of the bound h): the proof charges 6 u + 4 u |L|, the model needs far less
    h = f32(4.0 / 3.0) * U * (6.0 + 4.0 * np.abs(L.astype(np.float64)))
    assert ((e_star.astype(np.float64) - e_lo) / h).min() > 0.05
