"""The fp16 contrast path (BASELINE config 5; no reference counterpart, parity unpinned): weights dequantised and
rounded to binary16 at attach, fp32 activations.  Checked against oracle orc_forward_f16, which uses the same
binary16 weights and accumulates every product in double."""
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,steps", [("tiny", 40), ("small", 70), ("4Bmini", 10)])
def test_fp16_path_matches_double_precision_restatement(hip, host, orc, name, steps):
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    spec = Q.synth(name, path)
    mg = hip.q3_model_open(path.encode(), 0, 0)
    assert hip.q3_device_attach_fp16(mg) == 0
    mq = hip.q3_model_open(path.encode(), 0, 0)                 # the Q8_0 path, for contrast
    mo = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_threads(8)
    feed = np.random.default_rng(5).integers(0, spec.vocab_size, size=steps)
    worst, differs = 0.0, False
    for pos in range(min(steps, spec.seq_len)):
        lg = Q.logits_array(mg, hip.forward(mg, int(feed[pos]), pos))
        lo = Q.logits_array(mo, orc.orc_forward_f16(mo, int(feed[pos]), pos))
        lq = Q.logits_array(mq, hip.forward(mq, int(feed[pos]), pos))
        worst = max(worst, float(np.abs(lg - lo).max() / np.abs(lo).max()))
        differs = differs or not np.array_equal(lg, lq)
    # fp32 sums against double sums of identical binary16 weights: accumulation noise only
    assert worst <= 2e-4, worst
    assert differs                                              # it really is another arithmetic than the Q8_0 path
    hip.q3_model_close(mg); hip.q3_model_close(mq); host.q3_model_close(mo)
