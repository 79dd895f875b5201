/*
 * q3_model.c -- reader of the reference's `.bin` checkpoint layout.
 *
 * Written against the format, not against the reference loader: the layout is
 * defined by the exporter (reference qwen3/weights.py:249-381) and consumed by
 * reference src/model.c:59-244; SURVEY.md Appendix A.1 restates it:
 *
 *   [0,48)    12 x int32 ModelParams          [48,256) zero padding
 *   fp32      att_rms_norm[L][dim] ffn_rms_norm[L][dim] out_rms_norm[dim]
 *             q_rms_norm[L][hd] k_rms_norm[L][hd]
 *   Q8 tensor = int8[numel] followed by fp32[numel/64], in the order
 *             embedding, wq[L], wk[L], wv[L], wo[L], w1[L], w2[L], w3[L], (lm_head)
 *
 * The result is a `Model` with the reference's struct layout (q3_abi.h), so it
 * can be handed to forward() here or to code compiled against the reference
 * headers.  Unlike reference model_create() it does not dequantise the whole
 * embedding table or calloc a host KV cache unless Q3_OPEN_HOST_STATE is set:
 * the device path needs neither.
 */
#define _GNU_SOURCE
#include "q3_ext.h"

#include <fcntl.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct Q3Cursor {
    const uint8_t* base;
    size_t off;
    size_t size;
    int ok;
} Q3Cursor;

static const void* cursor_take(Q3Cursor* c, size_t bytes) {
    if (!c->ok || bytes > c->size - c->off) {
        c->ok = 0;
        return NULL;
    }
    const void* p = c->base + c->off;
    c->off += bytes;
    return p;
}

static Q8Tensor* carve_q8(Q3Cursor* c, int count, size_t numel, int group) {
    Q8Tensor* t = (Q8Tensor*)calloc((size_t)count, sizeof(Q8Tensor));
    if (!t) {
        c->ok = 0;
        return NULL;
    }
    for (int i = 0; i < count; i++) {
        t[i].q = (int8_t*)cursor_take(c, numel);
        t[i].s = (float*)cursor_take(c, (numel / (size_t)group) * sizeof(float));
    }
    return t;
}

static void* zalloc_pages(size_t bytes) {
    void* p = NULL;
    size_t rounded = (bytes + 4095) & ~(size_t)4095;
    if (posix_memalign(&p, 4096, rounded ? rounded : 4096) != 0) {
        return NULL;
    }
    memset(p, 0, rounded);
    return p;
}

static int params_sane(const ModelParams* p) {
    if (p->magic != Q3_MAGIC || p->version != Q3_VERSION) return 0;
    if (p->block_size != Q3_GROUP) return 0;
    if (p->dim <= 0 || p->hidden_dim <= 0 || p->n_layers <= 0 || p->n_heads <= 0) return 0;
    if (p->n_kv_heads <= 0 || p->vocab_size <= 0 || p->seq_len <= 0 || p->head_dim <= 0) return 0;
    if (p->n_heads % p->n_kv_heads) return 0;
    if (p->dim % Q3_GROUP || p->hidden_dim % Q3_GROUP) return 0;
    if ((p->n_heads * p->head_dim) % Q3_GROUP || (p->n_kv_heads * p->head_dim) % Q3_GROUP) return 0;
    if (p->head_dim % 8 || p->head_dim > 128) return 0;
    return 1;
}

static int alloc_host_state(Model* m) {
    const ModelParams* p = &m->params;
    ForwardState* s = &m->state;
    const size_t P = (size_t)p->n_heads * p->head_dim;
    const size_t kvd = (size_t)p->n_kv_heads * p->head_dim;
    const size_t big = P > (size_t)p->dim ? P : (size_t)p->dim;
    const size_t cache = (size_t)p->n_layers * p->seq_len * kvd;
    s->x = (float*)calloc((size_t)p->dim, sizeof(float));
    s->x_rms_norm = (float*)calloc(big, sizeof(float));
    s->q = (float*)calloc(P, sizeof(float));
    s->scores = (float*)calloc((size_t)p->n_heads * p->seq_len, sizeof(float));
    s->k_cache = (float*)calloc(cache, sizeof(float));
    s->v_cache = (float*)calloc(cache, sizeof(float));
    s->mlp_in = (float*)calloc((size_t)p->hidden_dim, sizeof(float));
    s->mlp_gate = (float*)calloc((size_t)p->hidden_dim, sizeof(float));
    s->qx.q = (int8_t*)calloc(big, 1);
    s->qx.s = (float*)calloc(big / Q3_GROUP, sizeof(float));
    s->qh.q = (int8_t*)calloc((size_t)p->hidden_dim, 1);
    s->qh.s = (float*)calloc((size_t)p->hidden_dim / Q3_GROUP, sizeof(float));
    if (!s->x || !s->x_rms_norm || !s->q || !s->scores || !s->k_cache || !s->v_cache
        || !s->mlp_in || !s->mlp_gate || !s->qx.q || !s->qx.s || !s->qh.q || !s->qh.s) {
        return 0;
    }
    /* fp32 copy of the embedding, q*s per element (reference src/q8.c:32-36) */
    const size_t ne = (size_t)p->vocab_size * p->dim;
    m->weights.fe = (float*)malloc(ne * sizeof(float));
    if (!m->weights.fe) return 0;
    const Q8Tensor* qe = m->weights.qe;
    for (size_t i = 0; i < ne; i++) {
        m->weights.fe[i] = (float)qe->q[i] * qe->s[i / Q3_GROUP];
    }
    return 1;
}

Model* q3_model_open(const char* path, int override_seq_len, int flags) {
    if (!path) return NULL;
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        fprintf(stderr, "[q3model] cannot open %s\n", path);
        return NULL;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < Q3_HEADER_BYTES) {
        fprintf(stderr, "[q3model] %s: too short for a checkpoint\n", path);
        close(fd);
        return NULL;
    }
    void* map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) {
        fprintf(stderr, "[q3model] mmap of %s failed\n", path);
        return NULL;
    }
    Model* m = (Model*)calloc(1, sizeof(Model));
    if (!m) {
        munmap(map, (size_t)st.st_size);
        return NULL;
    }
    m->data = map;
    m->size = st.st_size;
    memcpy(&m->params, map, sizeof(ModelParams));
    ModelParams* p = &m->params;
    if (!params_sane(p)) {
        fprintf(stderr, "[q3model] %s: bad header (magic %x version %d group %d)\n", path,
                p->magic, p->version, p->block_size);
        goto fail;
    }
    if (override_seq_len > 0 && override_seq_len <= p->seq_len) {
        p->seq_len = override_seq_len;
    }

    Q3Cursor c = { (const uint8_t*)map, Q3_HEADER_BYTES, (size_t)st.st_size, 1 };
    ModelWeights* w = &m->weights;
    const size_t L = (size_t)p->n_layers, dim = (size_t)p->dim, hd = (size_t)p->head_dim;
    const size_t P = (size_t)p->n_heads * hd, kvd = (size_t)p->n_kv_heads * hd;
    const size_t hid = (size_t)p->hidden_dim, V = (size_t)p->vocab_size;
    w->att_rms_norm = (float*)cursor_take(&c, L * dim * 4);
    w->ffn_rms_norm = (float*)cursor_take(&c, L * dim * 4);
    w->out_rms_norm = (float*)cursor_take(&c, dim * 4);
    w->q_rms_norm = (float*)cursor_take(&c, L * hd * 4);
    w->k_rms_norm = (float*)cursor_take(&c, L * hd * 4);
    w->qe = carve_q8(&c, 1, V * dim, Q3_GROUP);
    w->wq = carve_q8(&c, (int)L, P * dim, Q3_GROUP);
    w->wk = carve_q8(&c, (int)L, kvd * dim, Q3_GROUP);
    w->wv = carve_q8(&c, (int)L, kvd * dim, Q3_GROUP);
    w->wo = carve_q8(&c, (int)L, dim * P, Q3_GROUP);
    w->w1 = carve_q8(&c, (int)L, hid * dim, Q3_GROUP);
    w->w2 = carve_q8(&c, (int)L, dim * hid, Q3_GROUP);
    w->w3 = carve_q8(&c, (int)L, hid * dim, Q3_GROUP);
    w->cls = p->shared_classifier ? w->qe : carve_q8(&c, 1, V * dim, Q3_GROUP);
    if (!c.ok) {
        fprintf(stderr, "[q3model] %s: file is shorter than its header implies\n", path);
        goto fail;
    }
    m->state.logits = (float*)zalloc_pages(V * sizeof(float));
    if (!m->state.logits) goto fail;
    if ((flags & Q3_OPEN_HOST_STATE) && !alloc_host_state(m)) {
        fprintf(stderr, "[q3model] host state allocation failed\n");
        goto fail;
    }
    return m;

fail:
    q3_model_close(m);
    return NULL;
}

void q3_model_close(Model* m) {
    if (!m) return;
    extern void q3_device_detach(Model*) __attribute__((weak));
    if (q3_device_detach) q3_device_detach(m);
    ModelWeights* w = &m->weights;
    ForwardState* s = &m->state;
    if (w->cls != w->qe) free(w->cls);
    free(w->qe); free(w->wq); free(w->wk); free(w->wv); free(w->wo);
    free(w->w1); free(w->w2); free(w->w3); free(w->fe);
    free(s->x); free(s->x_rms_norm); free(s->q); free(s->scores); free(s->logits);
    free(s->k_cache); free(s->v_cache); free(s->mlp_in); free(s->mlp_gate);
    free(s->qx.q); free(s->qx.s); free(s->qh.q); free(s->qh.s);
    if (m->data) munmap(m->data, (size_t)m->size);
    free(m);
}

/* index of the first maximum (what `if (logits[i] > bv)` from the left yields).  Two SIMD passes --
 * the maximum, then the first 16-entry block that contains it -- instead of one branchy scalar walk
 * over 151,936 entries per token (220 us -> ~20 us on the bench host).  maxps(x, m) is exactly
 * `x > m ? x : m`, NaNs included. */
#if defined(__x86_64__)
/* the same two passes eight lanes wide where the CPU has AVX2 (picked at run time: the library is built for plain
 * x86-64): 24 us -> 9 us for 151,936 entries, which is host time between two steps of a greedy loop */
__attribute__((target("avx2"))) static int argmax_avx2(const float* logits, int n) {
    __m256 m0 = _mm256_set1_ps(logits[0]), m1 = m0, m2 = m0, m3 = m0;
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        m0 = _mm256_max_ps(_mm256_loadu_ps(logits + i), m0);
        m1 = _mm256_max_ps(_mm256_loadu_ps(logits + i + 8), m1);
        m2 = _mm256_max_ps(_mm256_loadu_ps(logits + i + 16), m2);
        m3 = _mm256_max_ps(_mm256_loadu_ps(logits + i + 24), m3);
    }
    float t[8];
    _mm256_storeu_ps(t, _mm256_max_ps(_mm256_max_ps(m0, m1), _mm256_max_ps(m2, m3)));
    float bv = t[0];
    for (int k = 1; k < 8; k++) bv = t[k] > bv ? t[k] : bv;
    for (; i < n; i++) bv = logits[i] > bv ? logits[i] : bv;
    const __m256 vb = _mm256_set1_ps(bv);
    int i0 = 0;
    for (; i0 + 32 <= n; i0 += 32) {
        const __m256 e = _mm256_or_ps(_mm256_or_ps(_mm256_cmp_ps(_mm256_loadu_ps(logits + i0), vb, _CMP_EQ_OQ),
                                                   _mm256_cmp_ps(_mm256_loadu_ps(logits + i0 + 8), vb, _CMP_EQ_OQ)),
                                      _mm256_or_ps(_mm256_cmp_ps(_mm256_loadu_ps(logits + i0 + 16), vb, _CMP_EQ_OQ),
                                                   _mm256_cmp_ps(_mm256_loadu_ps(logits + i0 + 24), vb, _CMP_EQ_OQ)));
        if (_mm256_movemask_ps(e)) break;
    }
    for (int k = i0; k < n; k++) {
        if (logits[k] == bv) return k;
    }
    return 0;
}
#endif

int q3_argmax(const float* logits, int n) {
#if defined(__x86_64__)
    if (n >= 64 && __builtin_cpu_supports("avx2")) return argmax_avx2(logits, n);
#endif
    float bv = logits[0];
    int i = 1;
#if defined(__SSE2__)
    if (n >= 64) {
        __m128 m0 = _mm_set1_ps(bv), m1 = m0, m2 = m0, m3 = m0;
        for (i = 0; i + 16 <= n; i += 16) {
            m0 = _mm_max_ps(_mm_loadu_ps(logits + i), m0);
            m1 = _mm_max_ps(_mm_loadu_ps(logits + i + 4), m1);
            m2 = _mm_max_ps(_mm_loadu_ps(logits + i + 8), m2);
            m3 = _mm_max_ps(_mm_loadu_ps(logits + i + 12), m3);
        }
        float t[4];
        _mm_storeu_ps(t, _mm_max_ps(_mm_max_ps(m0, m1), _mm_max_ps(m2, m3)));
        bv = t[0];
        for (int k = 1; k < 4; k++) bv = t[k] > bv ? t[k] : bv;
    }
#endif
    for (; i < n; i++) bv = logits[i] > bv ? logits[i] : bv;
    int i0 = 0;
#if defined(__SSE2__)
    const __m128 vb = _mm_set1_ps(bv);
    for (; i0 + 16 <= n; i0 += 16) {
        const __m128 e = _mm_or_ps(_mm_or_ps(_mm_cmpeq_ps(_mm_loadu_ps(logits + i0), vb), _mm_cmpeq_ps(_mm_loadu_ps(logits + i0 + 4), vb)),
                                   _mm_or_ps(_mm_cmpeq_ps(_mm_loadu_ps(logits + i0 + 8), vb), _mm_cmpeq_ps(_mm_loadu_ps(logits + i0 + 12), vb)));
        if (_mm_movemask_ps(e)) break;
    }
#endif
    for (int k = i0; k < n; k++) {
        if (logits[k] == bv) return k;
    }
    return 0;       /* only when every entry is NaN: the scalar walk would have kept index 0 too */
}

double q3_gemv_bytes(int d, int n) {
    return (double)d * (double)n * (1.0 + 4.0 / Q3_GROUP);
}

/* weights (int8 + scales) once, fp32 norm vectors,
 * fp32 KV read for T cached positions plus the write of one, one embedding row. */
double q3_bytes_per_token(const ModelParams* p, int T) {
    const double L = p->n_layers, dim = p->dim, hd = p->head_dim;
    const double P = (double)p->n_heads * hd, kvd = (double)p->n_kv_heads * hd;
    const double hid = p->hidden_dim, V = p->vocab_size;
    const double per_layer_elems = P * dim + 2.0 * kvd * dim + dim * P + 3.0 * hid * dim;
    const double mat = (L * per_layer_elems + V * dim) * (1.0 + 4.0 / Q3_GROUP);
    const double norms = 4.0 * (2.0 * L * dim + dim + 2.0 * L * hd);
    const double kv = 2.0 * L * kvd * 4.0 * ((double)T + 1.0);
    const double emb = dim * (1.0 + 4.0 / Q3_GROUP);
    return mat + norms + kv + emb;
}

/* ---- layer pipeline: who owns which layers, who works on what, when ------------------
 * Pure arithmetic, kept on the host side so that harnesses and CPU tests can use it
 * without loading the HIP runtime. */

/* Contiguous blocks of layers, sized so that the stages take equal TIME per tick: the last stage also
 * runs the classifier, which on MI355X streams at ~6.2 TB/s while a layer's five launches average
 * ~2.6 TB/s, so the classifier weighs 0.42 * cls_bytes / layer_bytes layers (1.6 layers at Qwen3-4B,
 * 67 us against 41 us per layer) and the last stage gets that many layers fewer.  With incomplete
 * parameters (or fewer layers than stages) the split is the plain even one. */
void q3_pipeline_layers(const ModelParams* p, int rank, int world, int* first, int* count) {
    const int L = p->n_layers;
    int cnt_last = -1;
    if (world > 1 && L >= world && p->dim > 0 && p->hidden_dim > 0 && p->vocab_size > 0 && p->n_heads > 0) {
        const double P = (double)p->n_heads * p->head_dim, KVD = (double)p->n_kv_heads * p->head_dim;
        const double layer = (double)p->dim * (P + 2.0 * KVD) + P * p->dim + 3.0 * (double)p->dim * p->hidden_dim;
        const double cls = 0.42 * (double)p->vocab_size * p->dim / layer;          /* in layers */
        const double target = ((double)L + cls) / world;
        cnt_last = (int)(target - cls + 0.5);
        if (cnt_last < 1) cnt_last = 1;
        if (cnt_last > L - (world - 1)) cnt_last = L - (world - 1);
    }
    if (cnt_last < 0) {
        const int base = L / world, rem = L % world;
        *count = base + (rank < rem ? 1 : 0);
        *first = rank * base + (rank < rem ? rank : rem);
        return;
    }
    const int rest = L - cnt_last, w1 = world - 1;
    const int base = rest / w1, rem = rest % w1;
    if (rank == world - 1) {
        *count = cnt_last;
        *first = rest;
    } else {
        *count = base + (rank < rem ? 1 : 0);
        *first = rank * base + (rank < rem ? rank : rem);
    }
}

/* `world` token streams circulate so that every stage is busy on every tick:
 * stream s, token k is on rank r at tick s + k*world + r. */
int q3_pipeline_schedule(int rank, int world, int nsteps, int tick, int* stream, int* k) {
    const int u = tick - rank;
    if (u < 0 || u >= nsteps * world) return 0;
    if (stream) *stream = u % world;
    if (k) *k = u / world;
    return 1;
}
