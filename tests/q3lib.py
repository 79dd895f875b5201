"""ctypes bindings shared by the tests, bench.py and __graft_entry__.py.

Three shared objects are involved:
  * qwen3.c_amd/libq3hip.so   -- the product (drop-in forward() + extensions)
  * qwen3.c_amd/libq3host.so  -- the product's plain-C host side alone
  * oracle/libq3oracle.so     -- the CPU checker (tests / smoke / cpu_baseline only)
  * oracle/_ref/libqwen3_ref.so -- the reference itself, built here from /root/reference
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "qwen3.c_amd")
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_float_p = C.POINTER(C.c_float)
c_int8_p = C.POINTER(C.c_int8)


class Q8Tensor(C.Structure):
    _fields_ = [("s", c_float_p), ("q", c_int8_p)]


class ModelParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "magic", "version", "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads",
        "vocab_size", "seq_len", "head_dim", "shared_classifier", "block_size")]


Q8P = C.POINTER(Q8Tensor)


class ModelWeights(C.Structure):
    _fields_ = [("wq", Q8P), ("wk", Q8P), ("wv", Q8P), ("wo", Q8P), ("w1", Q8P), ("w2", Q8P),
                ("w3", Q8P), ("cls", Q8P), ("qe", Q8P), ("fe", c_float_p),
                ("att_rms_norm", c_float_p), ("ffn_rms_norm", c_float_p),
                ("out_rms_norm", c_float_p), ("q_rms_norm", c_float_p),
                ("k_rms_norm", c_float_p)]


class ForwardState(C.Structure):
    _fields_ = [("x", c_float_p), ("x_rms_norm", c_float_p), ("q", c_float_p), ("k", c_float_p),
                ("v", c_float_p), ("scores", c_float_p), ("mlp_in", c_float_p),
                ("mlp_gate", c_float_p), ("logits", c_float_p), ("k_cache", c_float_p),
                ("v_cache", c_float_p), ("qx", Q8Tensor), ("qh", Q8Tensor)]


class Model(C.Structure):
    _fields_ = [("params", ModelParams), ("weights", ModelWeights), ("state", ForwardState),
                ("data", C.c_void_p), ("size", C.c_ssize_t)]


ModelP = C.POINTER(Model)


class SynthSpec(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len",
        "head_dim", "shared_classifier")] + [("seed", C.c_uint64), ("sigma", C.c_float)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("ms_total", C.c_double),
                ("bytes_per_launch", C.c_double)]


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def i8ptr(a):
    assert a.dtype == np.int8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int8_p)


def q8view(q, s):
    """Q8Tensor struct over two numpy arrays (kept alive by the caller)."""
    return Q8Tensor(fptr(s), i8ptr(q))


def _bind_host(lib):
    lib.q3_model_open.restype = ModelP
    lib.q3_model_open.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.q3_model_close.restype = None
    lib.q3_model_close.argtypes = [ModelP]
    lib.q3_synth_preset.restype = C.c_int
    lib.q3_synth_preset.argtypes = [C.c_char_p, C.POINTER(SynthSpec)]
    lib.q3_synth_write.restype = C.c_int
    lib.q3_synth_write.argtypes = [C.c_char_p, C.POINTER(SynthSpec)]
    lib.q3_synth_write_tokenizer.restype = C.c_int
    lib.q3_synth_write_tokenizer.argtypes = [C.c_char_p, C.c_int]
    lib.q3_synth_bytes.restype = C.c_int64
    lib.q3_synth_bytes.argtypes = [C.POINTER(SynthSpec)]
    lib.q3_file_checksum.restype = C.c_uint64
    lib.q3_file_checksum.argtypes = [C.c_char_p]
    lib.q3_argmax.restype = C.c_int
    lib.q3_argmax.argtypes = [c_float_p, C.c_int]
    lib.q3_bytes_per_token.restype = C.c_double
    lib.q3_bytes_per_token.argtypes = [C.POINTER(ModelParams), C.c_int]
    lib.q3_gemv_bytes.restype = C.c_double
    lib.q3_gemv_bytes.argtypes = [C.c_int, C.c_int]
    lib.q3_pipeline_layers.restype = None
    lib.q3_pipeline_layers.argtypes = [C.POINTER(ModelParams), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.q3_pipeline_schedule.restype = C.c_int
    lib.q3_pipeline_schedule.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return lib


_cache = {}


def build_all(quiet=True):
    """make the product libraries and the oracle (idempotent)."""
    out = None if not quiet else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", PKG, "all"], stdout=out)
    subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=out)


def host_lib():
    if "host" not in _cache:
        path = os.path.join(PKG, "libq3host.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", PKG, "host"], stdout=subprocess.DEVNULL)
        _cache["host"] = _bind_host(C.CDLL(path))
    return _cache["host"]


def hip_lib():
    """The product library.  Raises if it is missing: there is no fallback."""
    if "hip" not in _cache:
        path = os.environ.get("Q3_LIB") or os.path.join(PKG, "libq3hip.so")   # Q3_LIB: diagnostic variants
        if not os.path.exists(path):
            raise RuntimeError("libq3hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = _bind_host(C.CDLL(path))
        lib.forward.restype = c_float_p
        lib.forward.argtypes = [ModelP, C.c_int, C.c_int]
        lib.rmsnorm.restype = None
        lib.rmsnorm.argtypes = [c_float_p, c_float_p, c_float_p, C.c_int]
        lib.softmax.restype = None
        lib.softmax.argtypes = [c_float_p, C.c_int]
        lib.matmul.restype = None
        lib.matmul.argtypes = [c_float_p, Q8P, Q8P, C.c_int, C.c_int, C.c_int]
        lib.rotary.restype = None
        lib.rotary.argtypes = [c_float_p, C.c_int, C.c_int]
        lib.sigmoid.restype = C.c_float
        lib.sigmoid.argtypes = [C.c_float]
        lib.silu.restype = C.c_float
        lib.silu.argtypes = [C.c_float]
        lib.swiglu.restype = None
        lib.swiglu.argtypes = [c_float_p, c_float_p, C.c_int]
        lib.attention.restype = None
        lib.attention.argtypes = [ModelP, C.c_int, C.c_int]
        lib.q8_quantize.restype = None
        lib.q8_quantize.argtypes = [Q8P, c_float_p, C.c_int, C.c_int]
        lib.q8_dequantize.restype = None
        lib.q8_dequantize.argtypes = [Q8P, c_float_p, C.c_int, C.c_int]
        lib.q3_device_count.restype = C.c_int
        lib.q3_device_attach.restype = C.c_int
        lib.q3_device_attach.argtypes = [ModelP]
        lib.q3_device_detach.restype = None
        lib.q3_device_detach.argtypes = [ModelP]
        lib.q3_device_sync.restype = None
        lib.q3_device_sync.argtypes = [ModelP]
        lib.q3_forward_device.restype = None
        lib.q3_forward_device.argtypes = [ModelP, C.c_int, C.c_int]
        lib.q3_logits_fetch.restype = None
        lib.q3_logits_fetch.argtypes = [ModelP]
        lib.q3_handoff_fallbacks.restype = C.c_int
        lib.q3_handoff_fallbacks.argtypes = [ModelP]
        lib.q3_device_argmax.restype = C.c_int
        lib.q3_device_argmax.argtypes = [ModelP]
        lib.q3_generate_greedy.restype = C.c_int
        lib.q3_generate_greedy.argtypes = [ModelP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.q3_kv_fill_random.restype = None
        lib.q3_kv_fill_random.argtypes = [ModelP, C.c_int, C.c_uint64]
        lib.q3_tap_enable.restype = None
        lib.q3_tap_enable.argtypes = [ModelP, C.c_int]
        lib.q3_tap_data.restype = c_float_p
        lib.q3_tap_data.argtypes = [ModelP]
        lib.q3_layer_step.restype = None
        lib.q3_layer_step.argtypes = [ModelP, C.c_int, C.c_int, c_float_p, c_float_p]
        lib.q3_op_quantize.restype = None
        lib.q3_op_quantize.argtypes = [c_float_p, C.c_int, c_int8_p, c_float_p]
        lib.q3_op_rmsnorm_quantize.restype = None
        lib.q3_op_rmsnorm_quantize.argtypes = [c_float_p, c_float_p, C.c_int, c_float_p, c_int8_p, c_float_p]
        lib.q3_op_gemv.restype = None
        lib.q3_op_gemv.argtypes = [c_int8_p, c_float_p, c_int8_p, c_float_p, C.c_int, C.c_int, c_float_p]
        lib.q3_op_headnorm_rope.restype = None
        lib.q3_op_headnorm_rope.argtypes = [c_float_p, C.c_int, C.c_int, c_float_p, C.c_int]
        lib.q3_op_attention.restype = None
        lib.q3_op_attention.argtypes = [c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]
        lib.q3_op_swiglu.restype = None
        lib.q3_op_swiglu.argtypes = [c_float_p, c_float_p, C.c_int, c_float_p]
        lib.q3_op_expf.restype = None
        lib.q3_op_expf.argtypes = [c_float_p, C.c_int, c_float_p]
        lib.q3_prof_enable.restype = None
        lib.q3_prof_enable.argtypes = [ModelP, C.c_int]
        lib.q3_prof_reset.restype = None
        lib.q3_prof_reset.argtypes = [ModelP]
        lib.q3_prof_device_us.restype = C.c_double
        lib.q3_prof_device_us.argtypes = [ModelP, C.c_char_p]
        lib.q3_prof_get.restype = C.c_int
        lib.q3_prof_get.argtypes = [ModelP, C.POINTER(ProfEntry), C.c_int]
        lib.q3_pipeline_unique_id.restype = C.c_int
        lib.q3_pipeline_unique_id.argtypes = [C.c_void_p]
        lib.q3_pipeline_init.restype = C.c_int
        lib.q3_pipeline_init.argtypes = [C.c_int, C.c_int, C.c_void_p]
        lib.q3_pipeline_size.restype = C.c_int
        lib.q3_pipeline_layers.restype = None
        lib.q3_pipeline_layers.argtypes = [C.POINTER(ModelParams), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.q3_pipeline_schedule.restype = C.c_int
        lib.q3_pipeline_schedule.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.q3_pipeline_run.restype = C.c_int
        lib.q3_pipeline_run.argtypes = [ModelP, C.c_int, C.c_int, C.c_int]
        lib.q3_pipeline_run_streams.restype = C.c_int
        lib.q3_pipeline_run_streams.argtypes = [ModelP, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.q3_pipeline_selftest_streams.restype = C.c_int
        lib.q3_pipeline_selftest_streams.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.q3_pipeline_tokens.restype = C.c_int
        lib.q3_pipeline_tokens.argtypes = [ModelP, C.c_int, C.POINTER(C.c_int), C.c_int]
        lib.q3_pipeline_selftest.restype = C.c_int
        lib.q3_pipeline_selftest.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.q3_pipeline_allreduce_max.restype = C.c_double
        lib.q3_pipeline_allreduce_max.argtypes = [C.c_double]
        lib.q3_measure_copy_gbps.restype = C.c_double
        lib.q3_measure_copy_gbps.argtypes = [C.c_size_t, C.c_int]
        lib.q3_device_attach_fp16.restype = C.c_int
        lib.q3_device_attach_fp16.argtypes = [ModelP]
        lib.q3_complete.restype = C.c_int
        lib.q3_complete.argtypes = [ModelP, C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        lib.q3_prefill.restype = c_float_p
        lib.q3_prefill.argtypes = [ModelP, C.POINTER(C.c_int), C.c_int, C.c_int]
        lib.q3_op_gemm.restype = None
        lib.q3_op_gemm.argtypes = [c_int8_p, c_float_p, c_int8_p, c_float_p, C.c_int, C.c_int, C.c_int, c_float_p]
        lib.q3_op_sample.restype = C.c_int
        lib.q3_op_sample.argtypes = [c_float_p, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64)]
        lib.q3_device_sample.restype = C.c_int
        lib.q3_device_sample.argtypes = [ModelP, C.c_float, C.c_float, C.POINTER(C.c_uint64)]
        lib.q3_generate_sampled.restype = C.c_int
        lib.q3_generate_sampled.argtypes = [ModelP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        lib.q3_debug_gemv_loop.restype = C.c_double
        lib.q3_debug_gemv_loop.argtypes = [ModelP, C.c_char_p, C.c_int, C.c_int, C.c_int]
        lib.q3_debug_gemm_loop.restype = C.c_double
        lib.q3_debug_gemm_loop.argtypes = [ModelP, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.q3_debug_stamps.restype = C.c_int
        lib.q3_debug_stamps.argtypes = [ModelP, C.POINTER(C.c_uint64), C.c_int]
        lib.q3_pipeline_shutdown.restype = None
        lib.q3_version.restype = C.c_char_p
        _cache["hip"] = lib
    return _cache["hip"]


ORC_REF, ORC_TREE = 0, 1


def oracle_lib():
    if "orc" not in _cache:
        path = os.path.join(ORACLE_DIR, "libq3oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libq3oracle.so"], stdout=subprocess.DEVNULL)
        lib = C.CDLL(path)
        lib.orc_set_mode.argtypes = [C.c_int]
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_set_tap.argtypes = [c_float_p]
        lib.orc_expf.restype = C.c_float
        lib.orc_expf.argtypes = [C.c_float]
        lib.orc_q8_quantize.argtypes = [Q8P, c_float_p, C.c_int, C.c_int]
        lib.orc_q8_dequantize.argtypes = [Q8P, c_float_p, C.c_int, C.c_int]
        lib.orc_rmsnorm.argtypes = [c_float_p, c_float_p, c_float_p, C.c_int]
        lib.orc_softmax.argtypes = [c_float_p, C.c_int]
        lib.orc_matmul.argtypes = [c_float_p, Q8P, Q8P, C.c_int, C.c_int, C.c_int]
        lib.orc_rope_table.argtypes = [C.c_int, C.c_int, c_float_p, c_float_p]
        lib.orc_rotary.argtypes = [c_float_p, C.c_int, C.c_int]
        lib.orc_sigmoid.restype = C.c_float
        lib.orc_sigmoid.argtypes = [C.c_float]
        lib.orc_silu.restype = C.c_float
        lib.orc_silu.argtypes = [C.c_float]
        lib.orc_swiglu.argtypes = [c_float_p, c_float_p, C.c_int]
        lib.orc_attention_raw.argtypes = [c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]
        lib.orc_attention.argtypes = [ModelP, C.c_int, C.c_int]
        lib.orc_forward.restype = c_float_p
        lib.orc_forward.argtypes = [ModelP, C.c_int, C.c_int]
        lib.orc_layer_step.argtypes = [ModelP, C.c_int, C.c_int, c_float_p, c_float_p]
        lib.orc_kv_fill_random.restype = None
        lib.orc_kv_fill_random.argtypes = [ModelP, C.c_int, C.c_uint64]
        lib.orc_forward_f16.restype = c_float_p
        lib.orc_forward_f16.argtypes = [ModelP, C.c_int, C.c_int]
        lib.orc_xorshift_float.restype = C.c_float
        lib.orc_xorshift_float.argtypes = [C.POINTER(C.c_uint64)]
        lib.orc_sampler_clamp.argtypes = [c_float_p, c_float_p]
        lib.orc_sample.restype = C.c_int
        lib.orc_sample.argtypes = [c_float_p, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64)]
        _cache["orc"] = lib
    return _cache["orc"]


class RefSampler(C.Structure):
    """reference include/sampler.h:21-27"""
    _fields_ = [("dist", C.c_void_p), ("seed", C.c_uint64), ("temperature", C.c_float), ("top_p", C.c_float),
                ("vocab_size", C.c_int)]


def reference_lib(fast=False):
    """The reference compiled from /root/reference by oracle/Makefile, or None
    when neither the tree nor a prebuilt oracle/_ref exists."""
    key = "ref_fast" if fast else "ref"
    if key not in _cache:
        name = "libqwen3_ref_fast.so" if fast else "libqwen3_ref.so"
        path = os.path.join(ORACLE_DIR, "_ref", name)
        if not os.path.exists(path):
            if os.path.exists("/root/reference/src/forward.c"):
                subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
            else:
                _cache[key] = None
                return None
        lib = C.CDLL(path)
        lib.model_create.restype = ModelP
        lib.model_create.argtypes = [C.c_char_p, C.c_int]
        lib.model_free.argtypes = [ModelP]
        lib.forward.restype = c_float_p
        lib.forward.argtypes = [ModelP, C.c_int, C.c_int]
        lib.matmul.argtypes = [c_float_p, Q8P, Q8P, C.c_int, C.c_int, C.c_int]
        lib.rmsnorm.argtypes = [c_float_p, c_float_p, c_float_p, C.c_int]
        lib.softmax.argtypes = [c_float_p, C.c_int]
        lib.rotary.argtypes = [c_float_p, C.c_int, C.c_int]
        lib.swiglu.argtypes = [c_float_p, c_float_p, C.c_int]
        lib.attention.argtypes = [ModelP, C.c_int, C.c_int]
        lib.q8_quantize.argtypes = [Q8P, c_float_p, C.c_int, C.c_int]
        lib.sampler_create.restype = C.POINTER(RefSampler)
        lib.sampler_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint64]
        lib.sampler_free.argtypes = [C.POINTER(RefSampler)]
        lib.sample.restype = C.c_int
        lib.sample.argtypes = [C.POINTER(RefSampler), c_float_p]
        _cache[key] = lib
    return _cache[key]


def dropin_lib(twin=False):
    """The reference's library with src/forward.c and src/q8.c left out, linked against libq3hip.so
    (oracle/Makefile): the drop-in as a reference user would run it.  `twin`: the same reference sources
    linked against the CPU oracle's tree-order forward / softmax instead.  None when not built."""
    key = "dropin_twin" if twin else "dropin"
    if key not in _cache:
        path = os.path.join(ORACLE_DIR, "_ref", "libqwen3_dropin_oracle.so" if twin else "libqwen3_dropin.so")
        if not os.path.exists(path):
            if os.path.exists("/root/reference/src/forward.c") and os.path.exists(os.path.join(PKG, "libq3hip.so")):
                subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
            else:
                _cache[key] = None
                return None
        lib = C.CDLL(path)
        lib.model_create.restype = ModelP
        lib.model_create.argtypes = [C.c_char_p, C.c_int]
        lib.model_free.argtypes = [ModelP]
        lib.forward.restype = c_float_p            # resolved through the library's own dependency (libq3hip.so / the shim)
        lib.forward.argtypes = [ModelP, C.c_int, C.c_int]
        lib.sampler_create.restype = C.POINTER(RefSampler)
        lib.sampler_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint64]
        lib.sampler_free.argtypes = [C.POINTER(RefSampler)]
        lib.sample.restype = C.c_int
        lib.sample.argtypes = [C.POINTER(RefSampler), c_float_p]
        _cache[key] = lib
    return _cache[key]


def synth(name, path, seed=None, **overrides):
    """Write the named synthetic checkpoint to `path` (skipped if it already
    exists with the right size) and return the spec."""
    lib = host_lib()
    spec = SynthSpec()
    assert lib.q3_synth_preset(name.encode(), C.byref(spec)) == 0, name
    if seed is not None:
        spec.seed = seed
    for k, v in overrides.items():
        setattr(spec, k, v)
    want = lib.q3_synth_bytes(C.byref(spec))
    # the cached file is reused only for the very same spec (size alone does not tell seeds apart)
    key = repr([(n, getattr(spec, n)) for n, _ in SynthSpec._fields_])
    side = path + ".spec"
    same = os.path.exists(side) and open(side).read() == key
    if not (same and os.path.exists(path) and os.path.getsize(path) == want):
        if os.path.exists(side):
            os.remove(side)
        assert lib.q3_synth_write(path.encode(), C.byref(spec)) == 0
        with open(side, "w") as f:
            f.write(key)
    return spec


def logits_array(model_p, lib_logits=None):
    n = model_p.contents.params.vocab_size
    ptr = lib_logits if lib_logits is not None else model_p.contents.state.logits
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def tmp_dir():
    d = os.environ.get("Q3_TMP", "/tmp/q3")
    os.makedirs(d, exist_ok=True)
    return d


def record_parity(key, values):
    """Merge one entry of measured parity figures into the JSON the GPU runs leave under
    gpurun_out/ (copied to profiles/parity_rNN.json, tracked, after the run)."""
    import json
    path = os.environ.get("Q3_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    doc[key] = values
    with open(path + ".tmp", "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    os.replace(path + ".tmp", path)
