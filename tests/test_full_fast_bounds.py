"""CPU checks of the constants behind the full variant's fast path (gate_full.hip: full_pw_fast_kernel,
full_dw_fast_kernel): the float32 table GELU stays inside the error the kernels' bound tau assumes,
|gelu_f32(z) - gelu(z)| <= 4e-7 (|z| + 0.1) for the nearest-node form (gelu_node: the depthwise kernel) and
<= 1.6e-6 + 2e-7 |z| for the 1x1 kernel's tangent-line table (gelu_lin_node: 4096 nodes of width 1/256, nearest node).
The kernels' arithmetic is restated here in numpy float32 (same tables, entries rounded to float32; same operation
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


def _gelu_table(scale: float):
    """gelu_table_to_lds: 4096 nodes of width 1/256, the tangent at each node as a line in u = 256 z:
    (intercept, slope), intercept formed with the ROUNDED slope."""
    i = np.arange(4096) - 2048
    z = i / 256.0
    phi = np.array([0.5 * (1.0 + math.erf(v / math.sqrt(2.0))) for v in z])
    pdf = np.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
    slope = (scale * (phi + z * pdf) / 256.0).astype(np.float32)
    icpt = (scale * z * phi - i * slope.astype(np.float64)).astype(np.float32)
    icpt[0], slope[0] = 0.0, 0.0
    icpt[4095], slope[4095] = 0.0, np.float32(scale / 256.0)
    return icpt, slope


def _gelu_lin_f32(u, scale):
    """gelu_lin_node + the fma of full_pw_fast_kernel: u = the BatchNorm output in node widths (float32);
    t = med3(u + magic, lo, hi) -- a float32 add, which rounds u to the nearest integer --, entry (t - magic) + 2048,
    g = fma(u, slope, intercept)."""
    icpt, slope = _gelu_table(scale)
    magic = np.float32(12582912.0)
    t = (u.astype(np.float32) + magic).astype(np.float32)
    t = np.minimum(np.maximum(t, magic - np.float32(2048)), magic + np.float32(2047))
    k = (t - magic).astype(np.int64) + 2048
    return _fma32(u.astype(np.float32), slope[k], icpt[k]).astype(np.float64) / scale, k - 2048


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


def test_linear_table_gelu_error_bound():
    """The 1x1 kernel's GELU (round 3): tangent at the nearest of 4096 nodes; eg_m budgets 1.6e-6 + 2e-7 |z| for it
    (ADVICE, round 2: the node is now the NEAREST one -- the add of the magic number rounds u itself)."""
    rng = np.random.default_rng(11)
    z = np.concatenate([np.linspace(-12, 12, 800001), rng.normal(0, 2, 400000), rng.uniform(-9, 9, 200000),
                        (np.arange(-2048, 2048) + 0.5) / 256.0, (np.arange(-2048, 2048) + 0.5) / 256.0 + 1e-6])
    u = (256.0 * z).astype(np.float32)
    zf = u.astype(np.float64) / 256.0                 # the argument the kernel actually holds (its own error is ez_m's)
    exact = _gelu(zf)
    got, node = _gelu_lin_f32(u, 16.0)
    inside = np.abs(zf) < 7.99
    assert np.abs(node[inside] - u[inside].astype(np.float64)).max() <= 0.5          # the nearest node, always
    err = np.abs(got - exact)
    ratio = float((err / (1.6e-6 + 2e-7 * np.abs(zf))).max())
    assert 0.5 < ratio < 0.97, ratio                   # the interpolation term (1.53e-6 at z = 0) is nearly attained
    big = np.array([-50.0, -8.5, 8.5, 50.0])
    g, _ = _gelu_lin_f32((256.0 * big).astype(np.float32), 16.0)
    assert g[0] == 0.0 and g[1] == 0.0 and g[2] == 8.5 and g[3] == 50.0


def test_gelu_lower_bound_behind_the_signed_accumulator():
    """|g| <= g + 0.34: the kernel accumulates sum |w2| g and keeps 0.35 sum |w2| in the constant part of tau."""
    z = np.linspace(-10, 10, 2000001)
    assert _gelu(z).min() > -0.17


def test_table_edges_are_exact():
    for scale in (1.0, 16.0):
        big = np.array([-50.0, -8.5, 8.5, 50.0], dtype=np.float32)
        got = _gelu_f32(big, scale)
        assert got[0] == 0.0 and got[1] == 0.0 and got[2] == 8.5 and got[3] == 50.0
