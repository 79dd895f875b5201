"""The fp16 contrast path (BASELINE config 5; no reference counterpart, parity unpinned): weights dequantised and
rounded to binary16 at attach, fp32 activations.  Checked against oracle orc_forward_f16, which uses the same
binary16 weights and accumulates every product in double."""
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def _threads():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 8


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


@pytest.mark.parametrize("name,n", [("tiny", 40), ("small", 70), ("4Bmini", 80)])
def test_fp16_prefill_on_the_matrix_cores(hip, host, orc, name, n):
    """BASELINE config 5 in its batched form: q3_prefill of an fp16-attached model runs the four GEMMs of a layer
    on v_mfma_f32_16x16x32_f16 (binary16 weights x binary16-rounded activations, fp32 accumulate).  Parity
    unpinned (nothing in the reference does this); the checker is orc_forward_f16 fed token by token -- binary16
    weights, fp32 activations, double accumulation -- so the bar carries the activation rounding (2^-11 relative
    per element, averaged over the dot products and the layers)."""
    import ctypes as C
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    spec = Q.synth(name, path)
    n = min(n, spec.seq_len - 4)
    mg = hip.q3_model_open(path.encode(), 0, 0)
    assert hip.q3_device_attach_fp16(mg) == 0
    mo = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_threads(8)
    prompt = np.random.default_rng(17).integers(0, spec.vocab_size, size=n).astype(np.int32)
    arr = (C.c_int * n)(*[int(t) for t in prompt])
    lg = Q.logits_array(mg, hip.q3_prefill(mg, arr, n, 0))
    for pos in range(n):
        lo = Q.logits_array(mo, orc.orc_forward_f16(mo, int(prompt[pos]), pos))
    err_prefill = float(np.abs(lg - lo).max() / np.abs(lo).max())
    # a decode step on top of the cache the batched pass left
    tok = int(lo.argmax())
    lg2 = Q.logits_array(mg, hip.forward(mg, tok, n))
    lo2 = Q.logits_array(mo, orc.orc_forward_f16(mo, tok, n))
    err_next = float(np.abs(lg2 - lo2).max() / np.abs(lo2).max())
    orc.orc_set_threads(1)
    Q.record_parity(f"fp16_mfma_prefill_{name}", {"prompt": n, "rel_err_after_prompt": err_prefill, "rel_err_next_decode_step": err_next})
    # bar: 3e-3 (recorded errors are <= 8.5e-4, profiles/parity_r02.json; a dropped k-step or a wrong tile edge moves the
    # logits by percents, which the former 2e-2 bar would have let through -- ADVICE r2)
    assert np.isfinite(lg).all() and err_prefill <= 3e-3 and err_next <= 3e-3, (err_prefill, err_next)
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_fp16_path_full_size_4b(hip, host, orc):
    """BASELINE config 5 at its own size (Qwen3-4B shapes, 8.1 GB of binary16 weights): four teacher-forced decode
    steps against the double-accumulating checker at the 2e-4 bar of the small shapes, then a 64-token prompt through
    the MFMA GEMMs (q3_prefill on the fp16-attached model) and one decode step on the cache it leaves, at 3e-3.
    Parity unpinned by nature (nothing in the reference computes in binary16)."""
    import ctypes as C
    path = os.path.join(Q.tmp_dir(), "4B.bin")
    spec = Q.synth("4B", path)
    mg = hip.q3_model_open(path.encode(), 256, 0)
    assert hip.q3_device_attach_fp16(mg) == 0
    mo = host.q3_model_open(path.encode(), 256, 1)
    orc.orc_set_threads(_threads())
    feed = np.random.default_rng(11).integers(0, spec.vocab_size, size=4)
    worst = 0.0
    for pos in range(4):
        lg = Q.logits_array(mg, hip.forward(mg, int(feed[pos]), pos))
        lo = Q.logits_array(mo, orc.orc_forward_f16(mo, int(feed[pos]), pos))
        worst = max(worst, float(np.abs(lg - lo).max() / np.abs(lo).max()))
        assert int(lg.argmax()) == int(lo.argmax())
    assert worst <= 2e-4, worst
    hip.q3_model_close(mg); host.q3_model_close(mo)
    # the batched form on a fresh pair of models (position 0 again)
    n = 64
    mg = hip.q3_model_open(path.encode(), 256, 0)
    assert hip.q3_device_attach_fp16(mg) == 0
    mo = host.q3_model_open(path.encode(), 256, 1)
    prompt = np.random.default_rng(23).integers(0, spec.vocab_size, size=n).astype(np.int32)
    arr = (C.c_int * n)(*[int(t) for t in prompt])
    lg = Q.logits_array(mg, hip.q3_prefill(mg, arr, n, 0))
    for pos in range(n):
        lo = Q.logits_array(mo, orc.orc_forward_f16(mo, int(prompt[pos]), pos))
    err_prefill = float(np.abs(lg - lo).max() / np.abs(lo).max())
    tok = int(lo.argmax())
    lg2 = Q.logits_array(mg, hip.forward(mg, tok, n))
    lo2 = Q.logits_array(mo, orc.orc_forward_f16(mo, tok, n))
    err_next = float(np.abs(lg2 - lo2).max() / np.abs(lo2).max())
    orc.orc_set_threads(1)
    Q.record_parity("fp16_full_size_4B", {"decode_steps": 4, "rel_err_decode": worst, "prompt": n,
                                          "rel_err_after_prompt": err_prefill, "rel_err_next_decode_step": err_next})
    assert np.isfinite(lg).all() and err_prefill <= 3e-3 and err_next <= 3e-3, (err_prefill, err_next)
    hip.q3_model_close(mg); host.q3_model_close(mo)
