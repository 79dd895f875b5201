"""The C-ABI boundary without a GPU: struct layout identical to the reference's public
structs, every declared entry point exported, and a loud failure (never a CPU fallback)
when forward() is called on a machine without a usable device."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile

import q3lib as Q

HERE = os.path.dirname(os.path.abspath(__file__))


def test_struct_layout_matches_reference_headers():
    """golden/abi_layout.json was produced by compiling sizeof/offsetof against the
    reference's own include/model.h (tests/golden/make_golden.py)."""
    want = json.load(open(os.path.join(HERE, "golden", "abi_layout.json")))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "q3_abi.h"', 'int main(void){']
    for key in want:
        if "." in key:
            st, f = key.split(".")
            lines.append(f'printf("{key} %zu\\n", offsetof({st}, {f}));')
        else:
            lines.append(f'printf("{key} %zu\\n", sizeof({key}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "abi.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(td, "abi")
        subprocess.check_call(["gcc", "-I" + os.path.join(Q.ROOT, "include"), src, "-o", exe])
        got = {k: int(v) for k, v in (ln.split() for ln in subprocess.check_output([exe], text=True).splitlines())}
    assert got == want
    # the ctypes mirrors used by the tests agree as well
    assert C.sizeof(Q.Model) == want["Model"]
    assert Q.Model.state.offset == want["Model.state"]
    assert Q.ForwardState.logits.offset == want["ForwardState.logits"]


def declared_functions():
    names = set()
    for hdr in ("q3_forward.h", "q3_ext.h"):
        text = open(os.path.join(Q.ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert {"forward", "matmul", "rmsnorm", "softmax", "rotary", "sigmoid", "silu", "swiglu", "attention",
            "q8_quantize", "q8_dequantize", "q3_device_attach", "q3_pipeline_run"} <= names
    lib = C.CDLL(os.path.join(Q.PKG, "libq3hip.so"))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert Q.hip_lib().q3_version().startswith(b"q3hip")


def test_no_cpu_fallback_without_a_gpu():
    """On a box with no HIP device forward() must die with a message, not compute on the CPU."""
    hip = Q.hip_lib()
    if hip.q3_device_count() > 0:
        return      # a GPU is present (the GPU box runs the gpu-marked tests instead)
    path = os.path.join(Q.tmp_dir(), "tiny.bin")
    Q.synth("tiny", path)
    code = ("import sys; sys.path.insert(0, %r); import q3lib as Q; hip = Q.hip_lib();"
            "m = hip.q3_model_open(%r.encode(), 0, 0); hip.forward(m, 1, 0); print('COMPUTED')" % (HERE, path))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode != 0
    assert "COMPUTED" not in p.stdout
    assert "[q3hip]" in p.stderr and "no CPU path" in p.stderr


def test_reference_links_against_the_library_without_forward_c():
    """The drop-in claim of INTEGRATION.md section 1, at link time: the reference's other six
    translation units link into a complete shared object when src/forward.c and src/q8.c
    are replaced by -lq3hip (no undefined symbols allowed)."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "src", "forward.c")):
        import pytest
        pytest.skip("reference tree not present on this machine")
    srcs = [os.path.join(ref, "src", f + ".c") for f in ("xorshift", "tokenizer", "model", "sampler", "qwen", "completion")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "libqwen3_gpu.so")
        cmd = ["gcc", "-std=gnu17", "-DNDEBUG", "-O2", "-Wno-unused-result", "-I" + os.path.join(ref, "include"),
               "-fPIC", "-shared"] + srcs + ["-L" + Q.PKG, "-lq3hip", "-Wl,-rpath," + Q.PKG, "-Wl,--no-undefined",
                                             "-lm", "-o", out]
        subprocess.check_call(cmd)
        syms = subprocess.check_output(["nm", "-D", "--undefined-only", out], text=True)
        for name in ("forward", "softmax", "q8_dequantize"):
            assert re.search(r"\bU %s\b" % name, syms), name      # resolved from libq3hip.so
