/* dropin_oracle_shim.c -- test infrastructure only.
 *
 * The three symbols the reference's other translation units import from src/forward.c and src/q8.c
 * (forward, softmax, q8_dequantize: `nm -u` of them), answered by the CPU oracle in TREE order
 * (oracle/q3_oracle.c), i.e. by the arithmetic the GPU library is bit-identical to.  Linked with the
 * reference's loader / sampler / RNG it gives the CPU twin of oracle/_ref/libqwen3_dropin.so: the same
 * host code around a forward pass with the same bits, so the two must choose the same tokens for ANY
 * sampler setting (tests/test_gpu_dropin.py).  Never part of the product. */
#include "q3_abi.h"

void orc_set_mode(int mode);
float* orc_forward(Model* m, int token, int pos);
void orc_softmax(float* x, int size);
void orc_q8_dequantize(const Q8Tensor* qt, float* x, int n, int block_size);

float* forward(Model* m, int token, int pos) {
    orc_set_mode(1);
    return orc_forward(m, token, pos);
}

void softmax(float* x, int size) {
    orc_set_mode(1);
    orc_softmax(x, size);
}

void q8_dequantize(Q8Tensor* qt, float* x, int n, int block_size) { orc_q8_dequantize(qt, x, n, block_size); }
