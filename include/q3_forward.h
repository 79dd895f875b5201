/*
 * q3_forward.h -- the drop-in C-ABI: every symbol the reference's translation
 * unit src/forward.c (+ src/q8.c) exports, with identical names, argument
 * order and meaning, implemented by libq3hip.so on an MI355X.
 *
 * A maintainer replaces `src/forward.c src/q8.c` in the reference's library
 * target (CMakeLists.txt:28-37) by a link against libq3hip.so; see
 * INTEGRATION.md.  All pointers are HOST pointers, exactly as in the
 * reference; device residency is an implementation detail behind the call.
 *
 * Error behaviour follows the reference's convention (src/completion.c:27-30):
 * forward() has no error channel, so a HIP/RCCL failure prints a
 * "[q3hip] ..." line on stderr and calls exit(EXIT_FAILURE).  There is no CPU
 * fallback: without a usable GPU every entry point below fails that way.
 */
#ifndef Q3_FORWARD_H
#define Q3_FORWARD_H

#include "q3_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference include/forward.h:140 / src/forward.c:225-350.
 * One decode step: consumes `token` at position `pos` (KV entries 0..pos-1 of
 * every layer must come from earlier calls on the same Model), returns
 * m->state.logits (host float[vocab], fully rewritten on every call). */
float* forward(Model* m, int token, int pos);

/* reference include/forward.h:31 / src/forward.c:12-28.
 * out[i] = w[i] * (x[i] / sqrt(mean(x^2) + 1e-6)); out may alias x. */
void rmsnorm(float* out, float* x, float* w, int size);

/* reference include/forward.h:42 / src/forward.c:34-77. In place, any size
 * (the host sampler calls it on logits[vocab], src/sampler.c:196). */
void softmax(float* x, int size);

/* reference include/forward.h:61 / src/forward.c:79-101.
 * out[d] = W[d][n] . x[n], both Q8_0 with `block_size` groups along n. */
void matmul(float* out, Q8Tensor* x, Q8Tensor* w, int n, int d, int block_size);

/* reference include/forward.h:73 / src/forward.c:104-118. Half-split RoPE,
 * theta = 1e6, in place on one head. */
void rotary(float* x, int head_dim, int pos);

/* reference include/forward.h:85,93,107 / src/forward.c:122-139. */
float sigmoid(float x);
float silu(float x);
void swiglu(float* x1, float* x3, int size);

/* reference include/forward.h:124 / src/forward.c:141-195.
 * Attention of layer `layer` for the query at `pos` over cached positions
 * 0..pos, with the reference's HOST-state semantics: reads m->state.q (already
 * normed and rotated by the caller) and rows 0..pos of the layer's host
 * k_cache / v_cache, writes the head outputs to m->state.x_rms_norm.  It
 * neither norms, rotates nor appends, and does not touch the device-resident
 * cache forward() keeps.  A Model opened without host state (no q / KV cache
 * arrays) makes it print a message and exit: there is nothing to read. */
void attention(Model* m, int layer, int pos);

/* reference include/q8.h:25,30 / src/q8.c:5-37. */
void q8_quantize(Q8Tensor* qt, float* x, int n, int block_size);
void q8_dequantize(Q8Tensor* qt, float* x, int n, int block_size);

#ifdef __cplusplus
}
#endif
#endif /* Q3_FORWARD_H */
