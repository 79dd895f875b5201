"""Full-size parity (SURVEY.md 8(c') tiers C' and D) on the BASELINE.json shapes.

* 4B (headline config): GPU logits bit-identical to the oracle's tree order, run-to-run
  determinism, and the KV rewind semantics, at full width (2560 / 9728 / 151936).
* 0.6B (config 1, the reference's own CPU case): GPU vs the REAL reference build
  (oracle/_ref, 1 thread = golden), reported next to the reference's own self-noise
  (golden vs the same code at 16 threads): the reference does not reproduce itself to
  1e-3 at this size, so the bar is max(1e-3, 1.5 x self-noise), and every
  argmax disagreement must sit where the golden's top-2 gap is inside that noise."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def test_4b_logits_bit_exact_vs_tree_oracle(hip, host, orc):
    path = os.path.join(Q.tmp_dir(), "4B.bin")
    Q.synth("4B", path)
    mg = hip.q3_model_open(path.encode(), 128, 0)
    mo = host.q3_model_open(path.encode(), 128, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(16)
    feed = np.random.default_rng(8).integers(0, 151936, size=5)
    first = None
    for pos, tok in enumerate(feed):
        a = Q.logits_array(mg, hip.forward(mg, int(tok), pos))
        b = Q.logits_array(mo, orc.orc_forward(mo, int(tok), pos))
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), f"pos {pos}: max diff {np.abs(a - b).max()}"
        if pos == 0:
            first = a
    # rewind without clearing the cache (reference completion.c:281-284) reproduces position 0
    again = Q.logits_array(mg, hip.forward(mg, int(feed[0]), 0))
    assert np.array_equal(again, first)
    orc.orc_set_threads(1)
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_06b_against_the_reference_build_and_its_self_noise(hip):
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref not available")
    gomp = C.CDLL("libgomp.so.1")
    path = os.path.join(Q.tmp_dir(), "0.6B.bin")
    Q.synth("0.6B", path)
    feed = np.random.default_rng(9).integers(0, 151936, size=10)

    def run_ref(threads):
        gomp.omp_set_num_threads(threads)
        m = ref.model_create(path.encode(), 64)
        rows = [Q.logits_array(m, ref.forward(m, int(t), p)) for p, t in enumerate(feed)]
        ref.model_free(m)
        gomp.omp_set_num_threads(1)
        return np.stack(rows)

    golden = run_ref(1)
    noisy = run_ref(16)
    mg = hip.q3_model_open(path.encode(), 64, 0)
    gpu = np.stack([Q.logits_array(mg, hip.forward(mg, int(t), p)) for p, t in enumerate(feed)])
    hip.q3_model_close(mg)
    scale = np.abs(golden).max(axis=1)
    self_noise = float((np.abs(noisy - golden).max(axis=1) / scale).max())
    gpu_err = float((np.abs(gpu - golden).max(axis=1) / scale).max())
    print(f"0.6B: GPU vs golden {gpu_err:.3e}; reference 16 threads vs golden {self_noise:.3e}")
    assert gpu_err <= max(1e-3, 1.5 * self_noise)
    for p in range(len(feed)):
        if gpu[p].argmax() != golden[p].argmax():
            top2 = np.sort(golden[p])[-2:]
            assert (top2[1] - top2[0]) <= 2 * max(self_noise, gpu_err) * scale[p]
