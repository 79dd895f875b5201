"""The persistent step kernel (q3_mega.hip): one launch per decode step on Qwen3-4B layer
shapes.  Same parity bar as the multi-kernel path: logits bit-identical to the oracle's
tree order and to the multi-kernel path, across the 64-position chunk boundary, in
pipeline stages too."""
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


def open_with(hip, path, mega, seq=256):
    os.environ["Q3_MEGA"] = "1" if mega else "0"
    m = hip.q3_model_open(path.encode(), seq, 0)
    assert hip.q3_device_attach(m) == 0
    os.environ.pop("Q3_MEGA")
    assert hip.q3_uses_persistent_kernel(m) == (1 if mega else 0)
    return m


def test_mega_matches_multikernel_and_oracle(hip, host, orc):
    path = fixture_path()
    mm = open_with(hip, path, True)
    mk = open_with(hip, path, False)
    mo = host.q3_model_open(path.encode(), 256, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(8)
    feed = np.random.default_rng(5).integers(0, 8192, size=140)
    for pos, tok in enumerate(feed):
        a = Q.logits_array(mm, hip.forward(mm, int(tok), pos))
        b = Q.logits_array(mk, hip.forward(mk, int(tok), pos))
        assert np.array_equal(a, b), f"mega vs multi-kernel at pos {pos}"
        if pos < 6 or 60 <= pos < 70 or pos >= 136:
            c = Q.logits_array(mo, orc.orc_forward(mo, int(tok), pos))
            assert np.array_equal(a, c), f"mega vs oracle at pos {pos}"
        elif pos < 136:
            orc.orc_forward(mo, int(tok), pos)      # keep the oracle's KV cache in step
    orc.orc_set_threads(1)
    hip.q3_model_close(mm); hip.q3_model_close(mk); host.q3_model_close(mo)


def test_mega_greedy_loop_and_pipeline_stages(hip):
    path = fixture_path()
    os.environ["Q3_MEGA"] = "0"
    mk = hip.q3_model_open(path.encode(), 256, 0)
    n = 72
    want = (C.c_int * n)()
    assert hip.q3_generate_greedy(mk, 11, 0, n, want) == n
    hip.q3_model_close(mk)
    os.environ["Q3_MEGA"] = "1"
    mm = hip.q3_model_open(path.encode(), 256, 0)
    got = (C.c_int * n)()
    assert hip.q3_generate_greedy(mm, 11, 0, n, got) == n
    hip.q3_model_close(mm)
    assert list(got) == list(want)
    st = (C.c_int * (2 * n))()
    assert hip.q3_pipeline_selftest(path.encode(), 256, 2, 11, 0, n, st) == 0
    os.environ.pop("Q3_MEGA")
    assert list(st[:n]) == list(want) and list(st[n:]) == list(want)
