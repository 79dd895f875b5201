/*
 * q3_abi.h -- binary layout of the structs that cross the forward() boundary.
 *
 * The reference (teleprint-me/qwen3.c) passes a public, field-accessed `Model`
 * into forward(); callers read params.seq_len and state.logits directly
 * (reference src/completion.c:59,265).  A drop-in for src/forward.c therefore
 * has to agree with the reference on the *memory layout* of these five structs.
 * This header restates that layout (field order, types, names) -- it is an
 * interface mirror, nothing else:
 *
 *   Q8Tensor      <-> reference include/q8.h:16-19      (note: scales first)
 *   ModelParams   <-> reference include/model.h:30-43   (first 48 B of a .bin)
 *   ModelWeights  <-> reference include/model.h:55-82
 *   ForwardState  <-> reference include/model.h:92-117
 *   Model         <-> reference include/model.h:123-129
 *
 * tests/test_abi_and_symbols.py checks sizeof/offsetof of every field against the numbers
 * the reference headers produce with the same compiler.
 */
#ifndef Q3_ABI_H
#define Q3_ABI_H

#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3_MAGIC 0x7177656E   /* "qwen", reference include/model.h:15 */
#define Q3_VERSION 1
#define Q3_HEADER_BYTES 256   /* reference src/model.c:78-79 */
#define Q3_GROUP 64           /* the only group size the exporter emits (qwen3/__main__.py:22) */
#ifndef Q8_MAX
#define Q8_MAX 127.0f
#endif

/* A Q8_0 tensor view: one fp32 scale per `block_size` consecutive int8 codes. */
typedef struct Q8Tensor {
    float* s;
    int8_t* q;
} Q8Tensor;

/* Twelve little-endian int32 at file offset 0. */
typedef struct ModelParams {
    int magic;
    int version;
    int dim;               /* residual width                         */
    int hidden_dim;        /* FFN width                              */
    int n_layers;
    int n_heads;           /* query heads                            */
    int n_kv_heads;        /* key/value heads (GQA)                  */
    int vocab_size;
    int seq_len;           /* context length (may be overridden down)*/
    int head_dim;
    int shared_classifier; /* 1: lm_head is the embedding matrix     */
    int block_size;        /* quantisation group, 64                 */
} ModelParams;

typedef struct ModelWeights {
    Q8Tensor* wq;          /* [L] each (n_heads*head_dim) x dim      */
    Q8Tensor* wk;          /* [L] each (n_kv_heads*head_dim) x dim   */
    Q8Tensor* wv;          /* [L] each (n_kv_heads*head_dim) x dim   */
    Q8Tensor* wo;          /* [L] each dim x (n_heads*head_dim)      */
    Q8Tensor* w1;          /* [L] gate: hidden_dim x dim             */
    Q8Tensor* w2;          /* [L] down: dim x hidden_dim             */
    Q8Tensor* w3;          /* [L] up:   hidden_dim x dim             */
    Q8Tensor* cls;         /* vocab x dim (== qe when tied)          */
    Q8Tensor* qe;          /* quantised embedding, vocab x dim       */
    float* fe;             /* host fp32 embedding (reference only; may be NULL here) */
    float* att_rms_norm;   /* [L][dim]                               */
    float* ffn_rms_norm;   /* [L][dim]                               */
    float* out_rms_norm;   /* [dim]                                  */
    float* q_rms_norm;     /* [L][head_dim]                          */
    float* k_rms_norm;     /* [L][head_dim]                          */
} ModelWeights;

typedef struct ForwardState {
    float* x;
    float* x_rms_norm;
    float* q;
    float* k;
    float* v;
    float* scores;
    float* mlp_in;
    float* mlp_gate;
    float* logits;         /* [vocab] -- what forward() returns      */
    float* k_cache;
    float* v_cache;
    Q8Tensor qx;
    Q8Tensor qh;
} ForwardState;

typedef struct Model {
    ModelParams params;
    ModelWeights weights;
    ForwardState state;
    void* data;            /* base of the mmap                        */
    ssize_t size;          /* bytes mapped                            */
} Model;

#ifdef __cplusplus
}
#endif
#endif /* Q3_ABI_H */
