"""AddressSanitizer + UndefinedBehaviorSanitizer over the plain-C host side (loader, synthetic
writer) and the oracle, on the CPU build (GPU sanitizers are not available on the pool).  The
reference offers the same flags only in its Debug CMake configuration (CMakeLists.txt:13-16)."""
import os
import subprocess
import tempfile

import q3lib as Q

DRIVER = r'''
#include "q3_ext.h"
#include <stdio.h>
float* orc_forward(Model*, int, int);
void orc_set_mode(int); void orc_set_threads(int);
int main(int argc, char** argv) {
    const char* names[] = {"tiny", "small"};
    for (int k = 0; k < 2; k++) {
        Q3SynthSpec sp; q3_synth_preset(names[k], &sp);
        char path[512]; snprintf(path, sizeof path, "%s/%s.bin", argv[1], names[k]);
        if (q3_synth_write(path, &sp)) return 1;
        for (int mode = 0; mode < 2; mode++) {
            Model* m = q3_model_open(path, 0, Q3_OPEN_HOST_STATE);
            if (!m) return 2;
            orc_set_mode(mode); orc_set_threads(mode ? 4 : 1);
            int tok = 3;
            for (int pos = 0; pos < (k ? 70 : 64); pos++) {
                float* lg = orc_forward(m, tok, pos);
                tok = q3_argmax(lg, m->params.vocab_size);
            }
            q3_model_close(m);
        }
    }
    if (q3_model_open("/nonexistent/file.bin", 0, 0)) return 3;
    puts("SANITIZED-OK");
    return 0;
}
'''


def test_host_code_and_oracle_are_clean_under_asan_ubsan():
    inc = ["-I" + os.path.join(Q.ROOT, "include"), "-I" + os.path.join(Q.PKG, "csrc")]
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fopenmp"]
    srcs = [os.path.join(Q.PKG, "host", "q3_model.c"), os.path.join(Q.PKG, "host", "q3_synth.c"),
            os.path.join(Q.ORACLE_DIR, "q3_oracle.c")]
    with tempfile.TemporaryDirectory() as td:
        drv = os.path.join(td, "san.c")
        open(drv, "w").write(DRIVER)
        exe = os.path.join(td, "san")
        subprocess.check_call(["gcc", "-std=gnu17"] + san + inc + srcs + [drv, "-o", exe, "-lm"])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
        p = subprocess.run([exe, td], capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "SANITIZED-OK" in p.stdout
        assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
