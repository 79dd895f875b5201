/*
 * q3_synth.c -- deterministic random-init checkpoints in the reference's
 * `.bin` layout (format: reference qwen3/weights.py:249-381, read back by
 * reference src/model.c:59-244).
 *
 * There are no real Qwen3 weights offline, and BASELINE.json asks for
 * "random-init Qwen3-shaped weights"; the recipe follows SURVEY.md 8(d):
 * int8 codes uniform on [-127,127], one fp32 scale per 64 codes drawn as
 * (sigma/73.3)*U[0.75,1.25] (sigma scaled by 1/sqrt(2L) for wo and w2 so the
 * residual stays O(1) and greedy streams have usable top-1 gaps), RMSNorm
 * weights 1 + 0.1*N(0,1).  Every tensor has its own counter-based stream
 * (seed, tensor id, 4 KiB block) so the file can be produced in parallel and
 * is byte-identical on every machine; only integer ops and exactly rounded
 * float multiplies/adds are used (no libm).
 */
#define _GNU_SOURCE
#include "q3_ext.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct Rng {
    uint64_t key;
    uint64_t ctr;
} Rng;

static inline Rng rng_make(uint64_t seed, uint64_t tensor, uint64_t block) {
    Rng r;
    r.key = mix64(mix64(seed ^ 0x71337133ull) ^ mix64(tensor * 0x100000001B3ull + block));
    r.ctr = 0;
    return r;
}
static inline uint64_t rng_next(Rng* r) {
    return mix64(r->key + (r->ctr++) * 0x9E3779B97F4A7C15ull);
}
static inline float rng_unit(Rng* r) {            /* [0,1), 24 bits */
    return (float)(rng_next(r) >> 40) * (1.0f / 16777216.0f);
}
static inline float rng_normal(Rng* r) {          /* Irwin-Hall(12) - 6 */
    float acc = 0.0f;
    for (int i = 0; i < 6; i++) {
        uint64_t u = rng_next(r);
        acc += (float)(u >> 40) * (1.0f / 16777216.0f);
        acc += (float)((u >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
    }
    return acc - 6.0f;
}

enum { SEG_NORM = 0, SEG_CODES = 1, SEG_SCALES = 2 };

typedef struct Segment {
    int kind;
    uint64_t id;       /* stream id */
    int64_t offset;    /* file offset */
    int64_t count;     /* elements (floats or int8) */
    float sigma;       /* for SEG_SCALES */
} Segment;

#define BLOCK_ELEMS 4096

static void fill_block(const Segment* sg, uint64_t seed, int64_t blk, void* out, int n) {
    Rng r = rng_make(seed, sg->id, (uint64_t)blk);
    if (sg->kind == SEG_CODES) {
        int8_t* q = (int8_t*)out;
        int i = 0;
        while (i < n) {
            uint64_t u = rng_next(&r);
            for (int k = 0; k < 4 && i < n; k++, i++) {
                uint32_t h = (uint32_t)(u >> (16 * k)) & 0xFFFFu;
                q[i] = (int8_t)((int)((h * 255u) >> 16) - 127);
            }
        }
    } else if (sg->kind == SEG_SCALES) {
        float* s = (float*)out;
        const float unit = sg->sigma / 73.3f;
        for (int i = 0; i < n; i++) {
            s[i] = unit * (0.75f + 0.5f * rng_unit(&r));
        }
    } else {
        float* w = (float*)out;
        for (int i = 0; i < n; i++) {
            w[i] = 1.0f + 0.1f * rng_normal(&r);
        }
    }
}

static int write_all(int fd, const void* buf, size_t bytes, int64_t off) {
    const uint8_t* p = (const uint8_t*)buf;
    while (bytes) {
        ssize_t w = pwrite(fd, p, bytes, off);
        if (w < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        p += w;
        off += w;
        bytes -= (size_t)w;
    }
    return 0;
}

int q3_synth_preset(const char* name, Q3SynthSpec* s) {
    if (!name || !s) return -1;
    memset(s, 0, sizeof(*s));
    s->seed = 1234;
    s->sigma = 0.02f;
    s->vocab_size = 151936;
    s->seq_len = 40960;
    s->head_dim = 128;
    s->shared_classifier = 1;
    s->n_kv_heads = 8;
    if (!strcmp(name, "tiny")) {          /* the survey's fixture: 386,816 bytes */
        s->dim = 128; s->hidden_dim = 256; s->n_layers = 2; s->n_heads = 2; s->n_kv_heads = 1;
        s->head_dim = 64; s->vocab_size = 512; s->seq_len = 64;
    } else if (!strcmp(name, "small")) {  /* GQA 4:1, head_dim 128, ragged group counts */
        s->dim = 320; s->hidden_dim = 704; s->n_layers = 3; s->n_heads = 4; s->n_kv_heads = 1;
        s->head_dim = 128; s->vocab_size = 1024; s->seq_len = 512; s->shared_classifier = 0;
    } else if (!strcmp(name, "4Bmini")) { /* Qwen3-4B layer shapes, 2 layers, small vocabulary */
        s->dim = 2560; s->hidden_dim = 9728; s->n_layers = 2; s->n_heads = 32;
        s->vocab_size = 8192; s->seq_len = 1024;
    } else if (!strcmp(name, "0.6B")) {
        s->dim = 1024; s->hidden_dim = 3072; s->n_layers = 28; s->n_heads = 16;
    } else if (!strcmp(name, "1.7B")) {
        s->dim = 2048; s->hidden_dim = 6144; s->n_layers = 28; s->n_heads = 16;
    } else if (!strcmp(name, "4B")) {
        s->dim = 2560; s->hidden_dim = 9728; s->n_layers = 36; s->n_heads = 32;
    } else if (!strcmp(name, "8B")) {     /* DeepSeek-R1-0528-Qwen3-8B shapes, untied */
        s->dim = 4096; s->hidden_dim = 12288; s->n_layers = 36; s->n_heads = 32;
        s->shared_classifier = 0;
    } else {
        return -1;
    }
    return 0;
}

static int build_segments(const Q3SynthSpec* sp, Segment** out, int64_t* total) {
    const int64_t L = sp->n_layers, dim = sp->dim, hd = sp->head_dim;
    const int64_t P = (int64_t)sp->n_heads * hd, kvd = (int64_t)sp->n_kv_heads * hd;
    const int64_t hid = sp->hidden_dim, V = sp->vocab_size;
    const float sigma = sp->sigma > 0.0f ? sp->sigma : 0.02f;
    /* 1/sqrt(2L) without libm: Newton on y = 1/sqrt(a) */
    float a = 2.0f * (float)L, y = 1.0f / a;
    for (int i = 0; i < 40; i++) y = y * (1.5f - 0.5f * a * y * y);
    const float sigma_out = sigma * y;

    const int max_seg = 5 + 2 * (2 + 7 * (int)L);
    Segment* sg = (Segment*)calloc((size_t)max_seg, sizeof(Segment));
    if (!sg) return -1;
    int n = 0;
    int64_t off = Q3_HEADER_BYTES;
    uint64_t id = 1;
#define ADD_NORM(cnt) do { sg[n].kind = SEG_NORM; sg[n].id = id++; sg[n].offset = off; \
        sg[n].count = (cnt); off += 4 * (int64_t)(cnt); n++; } while (0)
#define ADD_Q8(numel, sig) do { \
        sg[n].kind = SEG_CODES; sg[n].id = id++; sg[n].offset = off; sg[n].count = (numel); \
        off += (int64_t)(numel); n++; \
        sg[n].kind = SEG_SCALES; sg[n].id = id++; sg[n].offset = off; sg[n].count = (numel) / Q3_GROUP; \
        sg[n].sigma = (sig); off += 4 * ((int64_t)(numel) / Q3_GROUP); n++; } while (0)
    ADD_NORM(L * dim);
    ADD_NORM(L * dim);
    ADD_NORM(dim);
    ADD_NORM(L * hd);
    ADD_NORM(L * hd);
    ADD_Q8(V * dim, sigma);
    for (int64_t l = 0; l < L; l++) ADD_Q8(P * dim, sigma);
    for (int64_t l = 0; l < L; l++) ADD_Q8(kvd * dim, sigma);
    for (int64_t l = 0; l < L; l++) ADD_Q8(kvd * dim, sigma);
    for (int64_t l = 0; l < L; l++) ADD_Q8(dim * P, sigma_out);
    for (int64_t l = 0; l < L; l++) ADD_Q8(hid * dim, sigma);
    for (int64_t l = 0; l < L; l++) ADD_Q8(dim * hid, sigma_out);
    for (int64_t l = 0; l < L; l++) ADD_Q8(hid * dim, sigma);
    if (!sp->shared_classifier) ADD_Q8(V * dim, sigma);
#undef ADD_NORM
#undef ADD_Q8
    *out = sg;
    *total = off;
    return n;
}

int64_t q3_synth_bytes(const Q3SynthSpec* spec) {
    Segment* sg = NULL;
    int64_t total = 0;
    if (!spec || build_segments(spec, &sg, &total) < 0) return -1;
    free(sg);
    return total;
}

int q3_synth_write(const char* path, const Q3SynthSpec* sp) {
    if (!path || !sp) return -1;
    if (sp->dim % Q3_GROUP || sp->hidden_dim % Q3_GROUP || (sp->n_heads * sp->head_dim) % Q3_GROUP
        || (sp->n_kv_heads * sp->head_dim) % Q3_GROUP || sp->n_heads % sp->n_kv_heads) {
        fprintf(stderr, "[q3synth] shape is not a multiple of the 64-wide group\n");
        return -1;
    }
    Segment* sg = NULL;
    int64_t total = 0;
    const int nseg = build_segments(sp, &sg, &total);
    if (nseg < 0) return -1;
    int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) {
        fprintf(stderr, "[q3synth] cannot create %s\n", path);
        free(sg);
        return -1;
    }
    if (ftruncate(fd, total) != 0) {
        close(fd);
        free(sg);
        return -1;
    }
    uint8_t header[Q3_HEADER_BYTES];
    memset(header, 0, sizeof(header));
    ModelParams hp;
    hp.magic = Q3_MAGIC; hp.version = Q3_VERSION; hp.dim = sp->dim; hp.hidden_dim = sp->hidden_dim;
    hp.n_layers = sp->n_layers; hp.n_heads = sp->n_heads; hp.n_kv_heads = sp->n_kv_heads;
    hp.vocab_size = sp->vocab_size; hp.seq_len = sp->seq_len; hp.head_dim = sp->head_dim;
    hp.shared_classifier = sp->shared_classifier ? 1 : 0; hp.block_size = Q3_GROUP;
    memcpy(header, &hp, sizeof(hp));
    int err = write_all(fd, header, sizeof(header), 0);

    /* flatten (segment, chunk) pairs so threads stay balanced on the big tensors */
    const int64_t CHUNK_BLOCKS = 256;  /* 256 * 4096 elements per work item */
    int64_t nitems = 0;
    for (int i = 0; i < nseg; i++) {
        int64_t blocks = (sg[i].count + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
        nitems += (blocks + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;
    }
    int64_t* item_seg = (int64_t*)malloc((size_t)nitems * sizeof(int64_t));
    int64_t* item_blk = (int64_t*)malloc((size_t)nitems * sizeof(int64_t));
    if (!item_seg || !item_blk) {
        free(item_seg); free(item_blk); free(sg); close(fd);
        return -1;
    }
    int64_t it = 0;
    for (int i = 0; i < nseg; i++) {
        int64_t blocks = (sg[i].count + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
        for (int64_t b = 0; b < blocks; b += CHUNK_BLOCKS) {
            item_seg[it] = i;
            item_blk[it] = b;
            it++;
        }
    }
#pragma omp parallel
    {
        uint8_t* buf = (uint8_t*)malloc((size_t)CHUNK_BLOCKS * BLOCK_ELEMS * 4);
#pragma omp for schedule(dynamic, 1)
        for (int64_t w = 0; w < nitems; w++) {
            const Segment* s = &sg[item_seg[w]];
            const int esz = (s->kind == SEG_CODES) ? 1 : 4;
            const int64_t blocks = (s->count + BLOCK_ELEMS - 1) / BLOCK_ELEMS;
            int64_t b0 = item_blk[w], b1 = b0 + CHUNK_BLOCKS;
            if (b1 > blocks) b1 = blocks;
            size_t filled = 0;
            for (int64_t b = b0; b < b1 && buf; b++) {
                int64_t left = s->count - b * BLOCK_ELEMS;
                int n = left < BLOCK_ELEMS ? (int)left : BLOCK_ELEMS;
                fill_block(s, sp->seed, b, buf + filled, n);
                filled += (size_t)n * (size_t)esz;
            }
            if (!buf || write_all(fd, buf, filled, s->offset + b0 * BLOCK_ELEMS * esz) != 0) {
#pragma omp atomic write
                err = -1;
            }
        }
        free(buf);
    }
    free(item_seg);
    free(item_blk);
    free(sg);
    if (close(fd) != 0) err = -1;
    if (err) fprintf(stderr, "[q3synth] write to %s failed\n", path);
    return err ? -1 : 0;
}

/* ---- synthetic tokenizer (.bin.tokenizer, version 2) ---------------------------------------------------
 * The reference's qwen_create() opens "<model path>.tokenizer" next to the checkpoint (src/qwen.c:14-49,
 * src/tokenizer.c:17-120; written by its exporter qwen3/tokenizer.py).  Layout, little endian:
 *   u32 magic "qtkn" 0x71746B6E | i32 version 2 | i32 vocab_size | i32 max_len | 10 x i32 special ids
 *   (bos eos eot pad bor eor btc etc btr etr) | per token: f32 score, i32 length, `length` bytes.
 * What is written: ids 1..255 the single bytes (id 0: a placeholder no input matches), a few merges so that the
 * reference's byte-pair loop has work to do, Qwen's nine special strings at the END of the vocabulary, and
 * unique printable fillers "[tNNNNNN]" everywhere else -- every id of a model's vocabulary decodes to text.
 * Same vocab_size => same bytes. */
static int tok_put(FILE* f, float score, const char* s, int len) {
    return fwrite(&score, 4, 1, f) == 1 && fwrite(&len, 4, 1, f) == 1 && (len == 0 || fwrite(s, 1, (size_t)len, f) == (size_t)len);
}
int q3_synth_write_tokenizer(const char* model_path, int vocab_size) {
    static const char* merges[] = {"he", "ll", "lo", "hello", " w", "or", "ld", " wor", " world", "th", "the", " the", "in", "er", "an", " a"};
    static const char* specials[] = {"<|endoftext|>", "<|im_end|>", "<|im_start|>", "<think>", "</think>", "<tool_call>", "</tool_call>",
                                     "<tool_response>", "</tool_response>"};
    const int n_merges = (int)(sizeof(merges) / sizeof(merges[0])), n_spec = (int)(sizeof(specials) / sizeof(specials[0]));
    if (!model_path || vocab_size < 256 + n_merges + n_spec) return -1;
    char* path = (char*)malloc(strlen(model_path) + 16);
    if (!path) return -1;
    sprintf(path, "%s.tokenizer", model_path);
    FILE* f = fopen(path, "wb");
    free(path);
    if (!f) return -1;
    const int spec0 = vocab_size - n_spec;               /* first special id */
    const uint32_t magic = 0x71746B6Eu;
    const int32_t head[3] = {2, vocab_size, 32};
    /* bos eos eot pad | bor eor | btc etc | btr etr   (pad = bos, as in Qwen's files) */
    const int32_t special[10] = {spec0, spec0 + 1, spec0 + 2, spec0, spec0 + 3, spec0 + 4, spec0 + 5, spec0 + 6, spec0 + 7, spec0 + 8};
    int ok = fwrite(&magic, 4, 1, f) == 1 && fwrite(head, 4, 3, f) == 3 && fwrite(special, 4, 10, f) == 10;
    for (int id = 0; ok && id < vocab_size; id++) {
        char buf[32];
        if (id == 0) {
            ok = tok_put(f, -1e9f, "[nul]", 5);
        } else if (id < 256) {
            buf[0] = (char)id;
            ok = tok_put(f, -1e6f, buf, 1);
        } else if (id < 256 + n_merges) {
            const char* m = merges[id - 256];
            ok = tok_put(f, (float)(n_merges - (id - 256)) + (float)strlen(m), m, (int)strlen(m));     /* longer merges win */
        } else if (id >= spec0) {
            const char* m = specials[id - spec0];
            ok = tok_put(f, 0.0f, m, (int)strlen(m));
        } else {
            const int n = snprintf(buf, sizeof(buf), "[t%06d]", id);
            ok = tok_put(f, -1e9f, buf, n);
        }
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}

uint64_t q3_file_checksum(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    uint64_t h = 0xcbf29ce484222325ull;
    static const size_t N = 1 << 20;
    uint8_t* buf = (uint8_t*)malloc(N);
    size_t got;
    while (buf && (got = fread(buf, 1, N, f)) > 0) {
        /* FNV-1a over 8-byte words (tail bytewise) keeps multi-GB files quick */
        size_t i = 0;
        for (; i + 8 <= got; i += 8) {
            uint64_t w;
            memcpy(&w, buf + i, 8);
            h = (h ^ w) * 0x100000001b3ull;
        }
        for (; i < got; i++) h = (h ^ buf[i]) * 0x100000001b3ull;
    }
    free(buf);
    fclose(f);
    return h;
}
