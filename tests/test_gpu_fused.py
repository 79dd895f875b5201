"""Stages fused across an in-launch hand-off (k_mlp: gate/up + SwiGLU -> h -> down in one
launch, the workgroups exchanging h through write-through stores and arrival counters).
Same parity bar as one launch per stage: logits bit-identical to the oracle's tree order and
to the unfused path, across the 64-position chunk boundary, through the graph replay, the
on-device greedy loop and the pipeline stages."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def fixture_path():
    path = os.path.join(Q.tmp_dir(), "4Bmini.bin")
    Q.synth("4Bmini", path)
    return path


MODES = {"plain": ("0", "0", 0), "mlp": ("1", "0", 1), "engine": ("1", "1", 2)}


def open_with(hip, path, mode, seq=256):
    fused, engine, want = MODES[mode]
    os.environ["Q3_FUSED"] = fused
    os.environ["Q3_ENGINE"] = engine
    m = hip.q3_model_open(path.encode(), seq, 0)
    assert hip.q3_device_attach(m) == 0
    os.environ.pop("Q3_FUSED"); os.environ.pop("Q3_ENGINE")
    assert hip.q3_fused_stages(m) == want, (mode, hip.q3_fused_stages(m))
    return m


@pytest.mark.parametrize("mode", ["mlp", "engine"])
def test_fused_matches_unfused_and_oracle(hip, host, orc, mode):
    path = fixture_path()
    mf = open_with(hip, path, mode)
    mk = open_with(hip, path, "plain")
    mo = host.q3_model_open(path.encode(), 256, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(8)
    feed = np.random.default_rng(5).integers(0, 8192, size=140)
    for pos, tok in enumerate(feed):
        a = Q.logits_array(mf, hip.forward(mf, int(tok), pos))
        b = Q.logits_array(mk, hip.forward(mk, int(tok), pos))
        assert np.array_equal(a, b), f"{mode} vs one launch per stage at pos {pos}: {np.abs(a - b).max()}"
        if pos < 6 or 60 <= pos < 70 or pos >= 136:
            c = Q.logits_array(mo, orc.orc_forward(mo, int(tok), pos))
            assert np.array_equal(a, c), f"{mode} vs oracle at pos {pos}"
        elif pos < 136:
            orc.orc_forward(mo, int(tok), pos)      # keep the oracle's KV cache in step
    orc.orc_set_threads(1)
    hip.q3_model_close(mf); hip.q3_model_close(mk); host.q3_model_close(mo)


@pytest.mark.parametrize("mode", ["mlp", "engine"])
def test_fused_layer_taps_match_oracle(hip, host, orc, mode):
    """the eager (non-graph) launch path of the fused kernels, residual after every layer"""
    path = fixture_path()
    mf = open_with(hip, path, mode)
    mo = host.q3_model_open(path.encode(), 256, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    L, dim = mf.contents.params.n_layers, mf.contents.params.dim
    tap = np.zeros((L, dim), dtype=np.float32)
    orc.orc_set_tap(Q.fptr(tap))
    hip.q3_tap_enable(mf, 1)
    for pos, tok in enumerate([17, 4000, 801]):
        hip.forward(mf, tok, pos)
        orc.orc_forward(mo, tok, pos)
        got = np.ctypeslib.as_array(hip.q3_tap_data(mf), shape=(L, dim)).copy()
        assert np.array_equal(got, tap), f"pos {pos}"
    orc.orc_set_tap(None)
    hip.q3_tap_enable(mf, 0)
    hip.q3_model_close(mf); host.q3_model_close(mo)


@pytest.mark.parametrize("mode", ["mlp", "engine"])
def test_fused_greedy_loop_and_pipeline_stages(hip, mode):
    path = fixture_path()
    n = 72
    mk = open_with(hip, path, "plain")
    want = (C.c_int * n)()
    assert hip.q3_generate_greedy(mk, 11, 0, n, want) == n
    hip.q3_model_close(mk)
    mf = open_with(hip, path, mode)
    got = (C.c_int * n)()
    assert hip.q3_generate_greedy(mf, 11, 0, n, got) == n
    hip.q3_model_close(mf)
    assert list(got) == list(want)
    st = (C.c_int * (2 * n))()
    os.environ["Q3_FUSED"], os.environ["Q3_ENGINE"] = MODES[mode][0], MODES[mode][1]
    try:
        assert hip.q3_pipeline_selftest(path.encode(), 256, 2, 11, 0, n, st) == 0
    finally:
        os.environ.pop("Q3_FUSED"); os.environ.pop("Q3_ENGINE")
    assert list(st[:n]) == list(want) and list(st[n:]) == list(want)
