"""Op-level parity (SURVEY.md 8(c') tier A): every kernel through the C-ABI against the
oracle on the same seeded inputs.

Bars: q8_quantize bit-exact with the reference order; every other op bit-exact with the
oracle's tree order (ORC_TREE restates the device reduction trees) AND within 1e-6 of
max|out| of the reference order (ORC_REF, pinned to the real reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu
REL = 1e-6


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rand_q8(rng, rows, n, sigma=0.02):
    q = rng.integers(-127, 128, size=(rows, n), dtype=np.int8)
    s = (sigma / 73.3 * rng.uniform(0.75, 1.25, size=(rows, n // 64))).astype(np.float32)
    return q, s


@pytest.mark.parametrize("n", [64, 128, 320, 2560, 4096, 9728])
def test_quantize_bit_exact(hip, orc, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * rng.uniform(0.01, 30)).astype(np.float32)
    x[: min(64, n)] = 0.0                       # an all-zero group takes the 1e-6 guard (q8.c:19)
    if n > 64:
        x[64] = 1e30                            # clamp path
        x[70] = 0.5 * x[65]                     # ties / half-way candidates
    q = np.zeros(n, np.int8); s = np.zeros(n // 64, np.float32)
    hip.q3_op_quantize(Q.fptr(x), n, Q.i8ptr(q), Q.fptr(s))
    q2 = np.zeros(n, np.int8); s2 = np.zeros(n // 64, np.float32)
    orc.orc_set_mode(Q.ORC_REF)
    t = Q.q8view(q2, s2)
    orc.orc_q8_quantize(C.byref(t), Q.fptr(x), n, 64)
    assert np.array_equal(s, s2)
    assert np.array_equal(q, q2)
    # the exported reference symbol goes through the same kernel
    q3 = np.zeros(n, np.int8); s3 = np.zeros(n // 64, np.float32)
    t3 = Q.q8view(q3, s3)
    hip.q8_quantize(C.byref(t3), Q.fptr(x), n, 64)
    assert np.array_equal(q3, q2) and np.array_equal(s3, s2)


@pytest.mark.parametrize("n,d", [(64, 2), (128, 256), (320, 768), (704, 320), (2560, 6144), (4096, 2560),
                                 (9728, 2560), (2560, 1024), (1024, 3072), (12288, 512), (5120, 64)])
def test_gemv(hip, orc, n, d):
    rng = np.random.default_rng(n * 7 + d)
    wq, ws = rand_q8(rng, d, n)
    xq, xs = rand_q8(rng, 1, n, sigma=1.0)
    xq, xs = xq[0].copy(), xs[0].copy()
    out = np.zeros(d, np.float32)
    hip.q3_op_gemv(Q.i8ptr(wq), Q.fptr(ws), Q.i8ptr(xq), Q.fptr(xs), n, d, Q.fptr(out))
    xt, wt = Q.q8view(xq, xs), Q.q8view(wq.reshape(-1), ws.reshape(-1))
    ref = np.zeros(d, np.float32); tree = np.zeros(d, np.float32)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_matmul(Q.fptr(tree), C.byref(xt), C.byref(wt), n, d, 64)
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_matmul(Q.fptr(ref), C.byref(xt), C.byref(wt), n, d, 64)
    assert np.array_equal(out, tree)
    assert rel_err(out, ref) <= REL
    out2 = np.zeros(d, np.float32)
    hip.matmul(Q.fptr(out2), C.byref(xt), C.byref(wt), n, d, 64)
    assert np.array_equal(out2, out)


@pytest.mark.parametrize("n", [64, 128, 320, 2560, 4096])
def test_rmsnorm_and_fused_quantize(hip, orc, n):
    rng = np.random.default_rng(n + 1)
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    out = np.zeros(n, np.float32)
    hip.rmsnorm(Q.fptr(out), Q.fptr(x), Q.fptr(w), n)
    tree = np.zeros(n, np.float32); ref = np.zeros(n, np.float32)
    orc.orc_set_mode(Q.ORC_TREE); orc.orc_rmsnorm(Q.fptr(tree), Q.fptr(x), Q.fptr(w), n)
    orc.orc_set_mode(Q.ORC_REF); orc.orc_rmsnorm(Q.fptr(ref), Q.fptr(x), Q.fptr(w), n)
    assert np.array_equal(out, tree)
    assert rel_err(out, ref) <= REL
    normed = np.zeros(n, np.float32); q = np.zeros(n, np.int8); s = np.zeros(n // 64, np.float32)
    hip.q3_op_rmsnorm_quantize(Q.fptr(x), Q.fptr(w), n, Q.fptr(normed), Q.i8ptr(q), Q.fptr(s))
    assert np.array_equal(normed, tree)
    q2 = np.zeros(n, np.int8); s2 = np.zeros(n // 64, np.float32); t = Q.q8view(q2, s2)
    orc.orc_q8_quantize(C.byref(t), Q.fptr(tree), n, 64)
    assert np.array_equal(q, q2) and np.array_equal(s, s2)


@pytest.mark.parametrize("hd,pos", [(128, 0), (128, 1), (128, 777), (128, 32767), (64, 5), (64, 63)])
def test_headnorm_rope(hip, orc, hd, pos):
    rng = np.random.default_rng(hd + pos)
    nh = 5
    heads = rng.standard_normal((nh, hd)).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    got = heads.copy()
    hip.q3_op_headnorm_rope(Q.fptr(got), nh, hd, Q.fptr(w), pos)
    for mode, exact in ((Q.ORC_TREE, True), (Q.ORC_REF, False)):
        orc.orc_set_mode(mode)
        exp = heads.copy()
        for h in range(nh):
            row = exp[h]
            orc.orc_rmsnorm(Q.fptr(row), Q.fptr(row), Q.fptr(w), hd)
            orc.orc_rotary(Q.fptr(row), hd, pos)
        if exact:
            assert np.array_equal(got, exp)
        else:
            assert rel_err(got, exp) <= REL
    # rotary alone (exported symbol): exact products with host libm cos/sin => bit-exact vs reference order
    one = heads[0].copy(); exp1 = heads[0].copy()
    hip.rotary(Q.fptr(one), hd, pos)
    orc.orc_rotary(Q.fptr(exp1), hd, pos)
    assert np.array_equal(one, exp1)


@pytest.mark.parametrize("T", [1, 2, 7, 63, 64, 65, 128, 130, 200, 1000, 1024, 1025, 2100, 4200, 8300])
@pytest.mark.parametrize("heads", [(4, 1, 128), (2, 1, 64), (8, 2, 128), (16, 8, 128)])
def test_attention(hip, orc, T, heads):
    H, KV, hd = heads
    rng = np.random.default_rng(T * 31 + H)
    q = rng.standard_normal((H, hd)).astype(np.float32)
    k = rng.standard_normal((T, KV, hd)).astype(np.float32)
    v = rng.standard_normal((T, KV, hd)).astype(np.float32)
    if T > 70:
        k[T // 2] *= 4.0          # a dominant key in a middle chunk exercises the chunk rescale
    got = np.zeros((H, hd), np.float32)
    hip.q3_op_attention(Q.fptr(q), Q.fptr(k), Q.fptr(v), T, H, KV, hd, Q.fptr(got))
    tree = np.zeros((H, hd), np.float32); ref = np.zeros((H, hd), np.float32)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_attention_raw(Q.fptr(q), Q.fptr(k), Q.fptr(v), T, H, KV, hd, Q.fptr(tree))
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_attention_raw(Q.fptr(q), Q.fptr(k), Q.fptr(v), T, H, KV, hd, Q.fptr(ref))
    assert np.array_equal(got, tree)
    # tree order vs the reference's sequential order: the gap is the reference's own summation noise,
    # which grows with the number of cached positions
    assert rel_err(got, ref) <= max(2e-6, 2e-9 * T)


def test_swiglu_expf_softmax_scalars(hip, orc):
    rng = np.random.default_rng(5)
    n = 9728
    g = (rng.standard_normal(n) * 4).astype(np.float32)
    u = rng.standard_normal(n).astype(np.float32)
    g[:6] = [0.0, -100.0, 100.0, -87.5, 88.0, 1e-30]
    got = np.zeros(n, np.float32)
    hip.q3_op_swiglu(Q.fptr(g), Q.fptr(u), n, Q.fptr(got))
    for mode, exact in ((Q.ORC_TREE, True), (Q.ORC_REF, False)):
        orc.orc_set_mode(mode)
        exp = g.copy()
        orc.orc_swiglu(Q.fptr(exp), Q.fptr(u), n)
        if exact:
            assert np.array_equal(got, exp)
        else:
            assert rel_err(got, exp) <= REL
    a = g.copy(); hip.swiglu(Q.fptr(a), Q.fptr(u), n)
    assert np.array_equal(a, got)
    # expf: device == host restatement bit for bit, and within 2 ulp of libm on the whole range
    x = np.concatenate([np.linspace(-90, 89, 200001), [0.0, -0.0, -86.0, -86.0001, 88.72283, 88.7229]]).astype(np.float32)
    e = np.zeros_like(x)
    hip.q3_op_expf(Q.fptr(x), len(x), Q.fptr(e))
    orc.orc_set_mode(Q.ORC_TREE)
    host = np.array([orc.orc_expf(float(t)) for t in x[::97]], np.float32)
    assert np.array_equal(e[::97], host)
    lib = np.exp(x.astype(np.float64))
    ok = (x >= -86) & (x <= 88.7)
    assert np.max(np.abs(e[ok] - lib[ok]) / lib[ok]) < 2.5e-7
    assert e[np.where(x == 0)[0][0]] == 1.0
    # scalar exports
    for t in (-3.0, 0.0, 0.7, 20.0):
        assert hip.sigmoid(t) == orc.orc_sigmoid(t)
        assert hip.silu(t) == orc.orc_silu(t)
    # softmax over a vocabulary-sized vector (the host sampler's call, sampler.c:196)
    for size in (1, 3, 512, 151936):
        z = (rng.standard_normal(size) * 5).astype(np.float32)
        got = z.copy(); hip.softmax(Q.fptr(got), size)
        orc.orc_set_mode(Q.ORC_TREE); t = z.copy(); orc.orc_softmax(Q.fptr(t), size)
        orc.orc_set_mode(Q.ORC_REF); r = z.copy(); orc.orc_softmax(Q.fptr(r), size)
        assert np.array_equal(got, t)
        # the reference sums `size` exponentials sequentially in fp32 (forward.c:55-68), which
        # itself drifts by ~size*2^-24; so the bar against it widens with size, and the
        # tree-summed result is additionally held to 1e-6 of the float64 answer
        assert rel_err(got, r) <= max(2e-6, 2e-9 * size)
        z64 = z.astype(np.float64); e64 = np.exp(z64 - z64.max()); e64 /= e64.sum()
        assert rel_err(got.astype(np.float64), e64) <= 1e-6


def test_dequantize(hip):
    rng = np.random.default_rng(9)
    n = 64 * 50
    q, s = rand_q8(rng, 1, n)
    q, s = q[0].copy(), s[0].copy()
    out = np.zeros(n, np.float32)
    t = Q.q8view(q, s)
    hip.q8_dequantize(C.byref(t), Q.fptr(out), n, 64)
    assert np.array_equal(out, q.astype(np.float32) * np.repeat(s, 64))


def _capture_stderr(fn):
    """Run fn() with file descriptor 2 redirected to a temporary file (C-level fprintf included); return the text."""
    import os, sys, tempfile
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile(mode="w+b") as tf:
        os.dup2(tf.fileno(), 2)
        try:
            fn()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tf.seek(0)
        return tf.read().decode()


@pytest.mark.parametrize("case", ["nan_first", "nan_mid", "plus_inf", "minus_inf", "finite"])
def test_softmax_export_keeps_the_reference_diagnostics(hip, orc, case):
    """The exported softmax() (host pointer in and out, arithmetic on the device) reports non-finite values the way
    the reference's does (src/forward.c:38-67) and leaves the same values behind: all NaN once a NaN or +Inf is in,
    exact zeros for -Inf."""
    rng = np.random.default_rng(3)
    n = 5000
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    if case == "nan_first": x[0] = np.nan
    if case == "nan_mid": x[1234] = np.nan
    if case == "plus_inf": x[77] = np.inf
    if case == "minus_inf": x[3] = -np.inf; x[4000] = -np.inf
    got = x.copy()
    text = _capture_stderr(lambda: hip.softmax(Q.fptr(got), n))
    ref = Q.reference_lib()
    if ref is not None:                       # the reference itself, built here from /root/reference (1 thread: its order)
        exp = x.copy()
        ref.softmax.restype = None
        ref.softmax.argtypes = [Q.c_float_p, C.c_int]
        rtext = _capture_stderr(lambda: ref.softmax(Q.fptr(exp), n))
        assert sorted(text.splitlines()) == sorted(rtext.splitlines())
        if os.environ.get("OMP_NUM_THREADS", "1") == "1":
            assert text == rtext
    else:
        orc.orc_set_mode(Q.ORC_REF)
        exp = x.copy()
        orc.orc_softmax(Q.fptr(exp), n)
    if case == "finite":
        assert text == ""
    if case == "nan_mid":
        assert "[Softmax] Invalid input: x[1234] = nan" in text and "[Softmax] NaN/Inf at i=1234: x=nan" in text
    if case == "nan_first":
        assert "Invalid input" not in text and text.count("[Softmax] NaN/Inf at i=") == n     # the reference never looks at x[0] in its first loop
    if case == "minus_inf":
        assert "[Softmax] Invalid input: x[3] = -inf" in text and "NaN/Inf at" not in text
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    if ok.any():
        # (fp32 sums in different orders: the device tree, the reference sequential or split over its OpenMP threads)
        assert float(np.abs(got[ok] - exp[ok]).max()) <= 5e-6 * float(np.abs(exp[ok]).max())
        assert np.array_equal(got[ok] == 0.0, exp[ok] == 0.0)
