"""CPU checks of the constants behind the full variant's fast path (gate_full.hip: full_pw_fast_kernel,
full_dw_fast_kernel): the float32 table GELU stays inside the error the kernels' bound tau assumes,
|gelu_f32(z) - gelu(z)| <= 4e-7 (|z| + 0.1).  The kernel's arithmetic is restated here in numpy float32
(same table: 512 nodes of width 1/32, value / slope / half curvature rounded to float32; same operation
order, fmas as float64 products rounded once)."""
import math

import numpy as np


def _phi_table(scale: float):
    z = (np.arange(512) - 256) / 32.0
    phi = np.array([0.5 * (1.0 + math.erf(v / math.sqrt(2.0))) for v in z])
    pdf = np.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
    c0, c1, c2 = phi.astype(np.float32), (pdf / scale).astype(np.float32), (-0.5 * z * pdf / scale ** 2).astype(np.float32)
    for edge, val in ((0, 0.0), (511, 1.0)):
        c0[edge], c1[edge], c2[edge] = val, 0.0, 0.0
    return c0, c1, c2


def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _gelu_f32(z, scale):
    c0, c1, c2 = _phi_table(scale)
    zs = (z * scale).astype(np.float32)                                   # the kernels carry z x scale
    r = np.rint((zs * np.float32(32.0 / scale)).astype(np.float32))
    r = np.minimum(np.maximum(r, np.float32(-256)), np.float32(255)).astype(np.float32)
    dz = _fma32(r, np.full_like(r, -scale / 32.0), zs)
    k = r.astype(np.int64) + 256
    p = _fma32(dz, _fma32(dz, c2[k], c1[k]), c0[k])
    return (zs * p).astype(np.float32).astype(np.float64) / scale


def _gelu(z):
    return np.array([0.5 * v * (1.0 + math.erf(v / math.sqrt(2.0))) for v in z])


def test_table_gelu_error_bound():
    rng = np.random.default_rng(7)
    z = np.concatenate([np.linspace(-12, 12, 200001), rng.normal(0, 2, 200000), rng.uniform(-9, 9, 100000)]).astype(np.float32)
    exact = _gelu(z.astype(np.float64))
    for scale in (1.0, 16.0):
        err = np.abs(_gelu_f32(z, scale) - exact)
        bound = 4e-7 * (np.abs(z.astype(np.float64)) + 0.1)
        assert (err <= bound).all(), (scale, float((err / bound).max()))
        assert float((err / bound).max()) < 0.8          # some margin left for the hardware's fused operations


def test_table_edges_are_exact():
    for scale in (1.0, 16.0):
        big = np.array([-50.0, -8.5, 8.5, 50.0], dtype=np.float32)
        got = _gelu_f32(big, scale)
        assert got[0] == 0.0 and got[1] == 0.0 and got[2] == 8.5 and got[3] == 50.0
