"""CPU checks of the constants behind the full variant's fast path (gate_full.hip: full_pw_fast_kernel,
full_dw_fast_kernel): the float32 table GELU stays inside the error the kernels' bound tau assumes,
|gelu_f32(z) - gelu(z)| <= 4e-7 (|z| + 0.1) for the nearest-node form (gelu_node: the depthwise kernel) and
<= 2.4e-6 (|z| + 0.1) for gelu_node_fast (the 1x1 kernel), whose node comes out of a float32 add of a shift that
is itself rounded to a whole node and can therefore be the second nearest.  The kernel's arithmetic is restated here in numpy float32
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


def _gelu_fast_f32(z, scale, shift):
    """gelu_node_fast as full_pw_fast_kernel runs it: z = d sc + shift (BatchNorm of the matrix output d);
    t = fma(d, 32 sc, float32(32 shift + magic)) -- the second operand rounded to an INTEGER at that magnitude --,
    node = t - magic, dz from the node actually taken."""
    c0, c1, c2 = _phi_table(scale)
    magic = np.float32(12582912.0)
    dsc = z.astype(np.float64) - shift
    sh2m = (32.0 * shift + 12582912.0).astype(np.float32)
    t = (32.0 * dsc + sh2m.astype(np.float64)).astype(np.float32)
    t = np.minimum(np.maximum(t, magic - np.float32(256)), magic + np.float32(255))
    zs = ((dsc + shift) * scale).astype(np.float32)
    r = (t - magic).astype(np.float32)
    dz = _fma32(r, np.full_like(r, -scale / 32.0), zs)
    k = r.astype(np.int64) + 256
    p = _fma32(dz, _fma32(dz, c2[k], c1[k]), c0[k])
    return (zs * p).astype(np.float32).astype(np.float64) / scale, r


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


def test_fast_node_selection_error_bound():
    """ADVICE (round 2): the shipped node selection can be one node off; its error is what eg_m budgets."""
    rng = np.random.default_rng(11)
    z = np.concatenate([np.linspace(-12, 12, 400001), rng.normal(0, 2, 400000), rng.uniform(-9, 9, 200000)])
    shift = rng.uniform(-3, 3, size=z.shape)
    exact = _gelu(z)
    got, node = _gelu_fast_f32(z, 16.0, shift)
    inside = np.abs(z) < 7.9
    assert np.abs(node[inside] / 32.0 - z[inside]).max() <= 1.0 / 32.0 + 1e-6          # never further than one node
    assert np.abs(node[inside] / 32.0 - z[inside]).max() > 1.0 / 64.0 + 1e-3           # ... and sometimes not the nearest
    err = np.abs(got - exact)
    assert float((err / (4e-7 * (np.abs(z) + 0.1))).max()) > 1.0                        # the nearest-node bound does NOT hold
    ratio = float((err / (2.4e-6 * (np.abs(z) + 0.1))).max())
    assert ratio < 0.7, ratio


def test_table_edges_are_exact():
    for scale in (1.0, 16.0):
        big = np.array([-50.0, -8.5, 8.5, 50.0], dtype=np.float32)
        got = _gelu_f32(big, scale)
        assert got[0] == 0.0 and got[1] == 0.0 and got[2] == 8.5 and got[3] == 50.0
