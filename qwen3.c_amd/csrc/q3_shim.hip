// q3_shim.hip -- the C-ABI of libq3hip.so: the reference's forward.h / q8.h symbols
// (include/q3_forward.h) and the extension entry points (include/q3_ext.h) on top of
// the gfx950 kernels in q3_kernels.hip.
//
// The reference's Model has no opaque slot for a backend (include/model.h:123-129) and
// forward() has no handle argument, so the device state of a Model lives in a side
// registry keyed by the Model pointer; it is created on the first forward() (weights
// are uploaded from the views the loader carved out of the mmap) and torn down by
// q3_device_detach()/q3_model_close() or at exit.
//
// HBM layout per Model (all hipMalloc'd once at attach, nothing allocated per step):
//   per layer   qkv  : rows of wq | wk | wv concatenated  [(P+2KVD)][dim] int8 + [..][dim/64] f32
//               wo   : [dim][P]            gate/up : rows interleaved g0,u0,g1,u1..  [2*HID][dim]
//               down : [dim][HID]          norms   : att, ffn [dim]; q, k [hd]
//               K, V : [n_kv][seq_len][hd] f32 (positions of one head contiguous, so one
//                      1 KiB wave-load is two whole key rows)
//   embedding / lm_head (shared when tied), final norm, RoPE table [seq_len][hd/2][2]
//   activations x[dim], qkv[P+2KVD], att codes[P]+scales, h[HID], logits[V], chunk partials
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "q3_ext.h"
#include "q3_forward.h"
#include "q3_kernels.hpp"
#include "q3_numerics.h"

#define Q3_DIE(...)                         \
    do {                                    \
        fprintf(stderr, "[q3hip] " __VA_ARGS__); \
        fprintf(stderr, "\n");              \
        exit(EXIT_FAILURE);                 \
    } while (0)

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            Q3_DIE("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace {

struct LayerDev {
    void *qkv_h = nullptr, *wo_h = nullptr, *gu_h = nullptr, *dn_h = nullptr;   // fp16 contrast path only (binary16 copies)
    int8_t *qkv_q, *wo_q, *gu_q, *dn_q;
    float *qkv_s, *wo_s, *gu_s, *dn_s;
    float *att_nw, *ffn_nw, *qnw, *knw;
    float *kc, *vc;
};

struct ProfSlot {
    const char* name;
    double bytes;
    int64_t launches;
    double ms;
    double dev_ms;      // same launches by the device clock inside the kernel (GEMV only)
    int64_t dev_launches;
};
constexpr int CLK_MAX = 512;

struct Pipe {
    bool on = false;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // Q3_PIPE_SELF=1 with world 1: a ONE-rank RCCL communicator whose rank sends the tick's message to itself -- the
    // transport of the pipeline (communicator set-up, grouped ncclSend/ncclRecv on the launch stream between graph
    // replays) exercised on the single GPU a test box has
    bool self = false;
};
Pipe g_pipe;

// What a registry entry was built from.  A Model freed by the reference's model_free() and a new one that
// malloc happens to put at the same address must not inherit the old device state.
struct ModelId {
    const void* data = nullptr;
    size_t size = 0;
    const void *wq0 = nullptr, *qe = nullptr, *logits = nullptr;
    int dim = 0, hid = 0, L = 0, V = 0, seq = 0;
    bool operator==(const ModelId& o) const {
        return data == o.data && size == o.size && wq0 == o.wq0 && qe == o.qe && logits == o.logits && dim == o.dim &&
               hid == o.hid && L == o.L && V == o.V && seq == o.seq;
    }
};
ModelId model_id(const Model* m) {
    ModelId id;
    id.data = m->data; id.size = (size_t)m->size;
    id.wq0 = (m->weights.wq && m->params.n_layers > 0) ? (const void*)m->weights.wq[0].q : nullptr;
    id.qe = m->weights.qe ? (const void*)m->weights.qe->q : nullptr;
    id.logits = m->state.logits;
    id.dim = m->params.dim; id.hid = m->params.hidden_dim; id.L = m->params.n_layers; id.V = m->params.vocab_size;
    id.seq = m->params.seq_len;
    return id;
}

struct Dev {
    Model* m = nullptr;
    ModelId id;
    float* logits_host = nullptr;  // m->state.logits at attach (registered with HIP when logits_pinned)
    int device = 0;
    hipStream_t st = nullptr;
    int dim = 0, hid = 0, L = 0, H = 0, KV = 0, hd = 0, P = 0, KVD = 0, V = 0, seq = 0;
    int seq_pad = 0;             // cache rows per kv head: seq rounded up to the 64-position chunk
    int l0 = 0, l1 = 0;          // layers [l0, l1) live on this device
    int rank = 0, world = 1;     // pipeline stage identity of this Dev
    bool loopback = false;       // self-test: all stages in one process, hand-offs by D2D copy
    Dev* loop_prev = nullptr;    // loopback: the stage that feeds this one (ring)
    float* outbox[2] = {nullptr, nullptr};   // loopback mailboxes, indexed by tick parity
    bool has_embed = true, has_cls = true;
    std::vector<LayerDev> layers; // indexed by global layer id; only [l0,l1) filled
    std::vector<void*> allocs;
    int8_t *emb_q = nullptr, *cls_q = nullptr, *att_q = nullptr;
    float *emb_s = nullptr, *cls_s = nullptr, *out_nw = nullptr;
    float *xout = nullptr;       // pipeline: outgoing message (residual + token slot)
    float *x = nullptr, *qkv = nullptr, *h = nullptr, *logits = nullptr, *att_s = nullptr;
    float *att_f = nullptr, *part = nullptr, *rope = nullptr, *tap_dev = nullptr, *cs_cur = nullptr;
    q3k::Ctl* ctl = nullptr;
    q3k::Ctl* ctl_host = nullptr;
    int* amax = nullptr;
    float* amax_scratch = nullptr;
    unsigned* tickets = nullptr;  // attention chunk tickets, [KV], zero between launches
    unsigned* epoch = nullptr;    // step counter advanced by k_begin: tags the in-launch hand-offs of k_attn_wo
    unsigned long long* att_g = nullptr;   // attention output as {tag, value} granules [P/4 + P/64] (fused launch)
    unsigned long long* part_g = nullptr;  // chunk partials as {tag, value} granules [H][max_chunks][hd+2] (k_attn_wo, in-launch merge)
    unsigned* err_host = nullptr; // host-mapped: [0] set by a consumer whose bounded wait gave up
    // What has been queued since the last synchronisation that found err_host clear, as closures that queue it again:
    // when a hand-off inside a fused launch times out, the Model drops to the separate launches (fuse = false, graphs
    // rebuilt) and the work is redone in this process (sync_checked).
    std::vector<std::function<void()>> redo;
    bool replaying = false;
    int fallbacks = 0;            // how many times that happened (q3_handoff_fallbacks)
    bool fuse = true;             // attention + Wo in one launch below Q3_ATT_LONG cached positions (Q3_FUSE=0: separate launches)
    int pf_delay = 150;           // x 10 ns: how long the consumer workgroups of k_attn_wo hold their Wo requests back (Q3_WO_DELAY)
    unsigned long long* stamps = nullptr;
    int* amax_host = nullptr;
    int max_chunks = 1, chunk_slots = 1;
    bool logits_pinned = false;
    bool use_graph = true;
    std::vector<hipGraphExec_t> gexec;    // one per launch shape of a step (q3k::step_shape)
    int nshapes = 5;
    // pipeline / on-device loop: one KV cache per concurrent token stream
    int n_streams = 1;
    size_t cache_floats = 0;      // floats of one stream's K (or V) cache of one layer
    std::vector<hipGraphExec_t> pgexec;   // [stream*nshapes + shape], step without the ctl upload
    int* ptokens = nullptr;       // last stage: [n_streams][cap] chosen tokens
    // fp16 contrast path (BASELINE config 5): weights dequantised to binary16 at attach, activations fp32
    bool fp16 = false;
    void *emb_h = nullptr, *cls_h = nullptr;
    // batched prompt ingestion (q3_prefill.hip): buffers for one chunk of 16 positions
    bool pf_ready = false;
    int* pf_tokens = nullptr;
    q3k::Ctl* pf_ctl = nullptr;
    float *pf_cs = nullptr, *pf_x = nullptr, *pf_qkv = nullptr, *pf_h = nullptr, *pf_as = nullptr, *pf_xs = nullptr;
    uint16_t* pf_xh = nullptr;    // fp16 attach: activation rows rounded to binary16 (MFMA operand)
    float* pf_attf = nullptr;     // fp16 attach: fp32 head outputs of the chunk [B][P]
    float* pf_part = nullptr;     // chunk partials of 16 positions
    unsigned* pf_tickets = nullptr;
    int8_t *pf_aq = nullptr, *pf_xq = nullptr;
    // device-side sampling (q3_sample.hip)
    q3k::SampleBufs sb = {};
    bool sb_ready = false;
    bool samp_on = false;         // the generation loop samples instead of taking the argmax
    float samp_t = 1.0f, samp_p = 1.0f;
    unsigned long long* seed_dev = nullptr;
    int* samp_tok = nullptr;      // device slot + pinned host copy of the sampled token
    int* samp_tok_host = nullptr;
    int ptokens_cap = 0;
    bool tap = false;
    std::vector<float> tap_host;
    bool prof = false;
    std::vector<ProfSlot> prof_slots;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int> ev_slot;     // slot index of each used pair in this step
    unsigned long long* clk_dev = nullptr;   // [2 * CLK_MAX] device-clock marks of the timed GEMV launches
    std::vector<unsigned long long> clk_host;
    size_t ev_used = 0;
};

std::mutex g_mu;
std::unordered_map<Model*, Dev*> g_reg;
bool g_atexit = false;

int pick_device() {
    const char* e = getenv("Q3_DEVICE");
    if (e && *e) return atoi(e);
    e = getenv("LOCAL_RANK");
    if (e && *e) {
        int n = 0;
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0) return atoi(e) % n;
    }
    return 0;
}

template <typename T>
T* dalloc(Dev* d, size_t count) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, count * sizeof(T) + 256));
    d->allocs.push_back(p);
    return reinterpret_cast<T*>(p);
}

template <typename T>
T* upload(Dev* d, const T* host, size_t count) {
    T* p = dalloc<T>(d, count);
    HIPCHK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

__global__ void k_interleave_rows(char* dst, const char* src, size_t row_bytes, size_t rows, int which) {
    // dst row 2r+which = src row r; 16-byte units
    const size_t units = row_bytes / 4;
    const size_t total = rows * units;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / units, u = i - r * units;
        reinterpret_cast<int*>(dst)[(2 * r + which) * units + u] = reinterpret_cast<const int*>(src)[i];
    }
}

void die_if_no_gpu() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        Q3_DIE("no usable HIP device (%s): this library has no CPU path",
               e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
}

void build_rope(Dev* d) {
    // the same libm calls, in the same order, as reference rotary() (forward.c:107-111)
    const int half = d->hd / 2;
    std::vector<float> tab((size_t)d->seq * d->hd);
#pragma omp parallel for schedule(static)
    for (int pos = 0; pos < d->seq; pos++) {
        for (int i = 0; i < half; i++) {
            const float angle = pos * powf(1e6f, -(float)i / half);
            tab[(size_t)pos * d->hd + 2 * i] = cosf(angle);
            tab[(size_t)pos * d->hd + 2 * i + 1] = sinf(angle);
        }
    }
    d->rope = upload<float>(d, tab.data(), tab.size());
}

void upload_weights(Dev* d) {
    const Model* m = d->m;
    const ModelWeights* w = &m->weights;
    const size_t dim = d->dim, P = d->P, KVD = d->KVD, hid = d->hid, V = d->V, hd = d->hd;
    const size_t G = Q3_GROUP;
    d->layers.resize(d->L);
    // staging buffer for the gate/up row interleave
    char* stage_q = nullptr;
    char* stage_s = nullptr;
    HIPCHK(hipMalloc((void**)&stage_q, hid * dim));
    HIPCHK(hipMalloc((void**)&stage_s, hid * dim / G * 4));
    for (int l = d->l0; l < d->l1; l++) {
        LayerDev& L = d->layers[l];
        const size_t rows = P + 2 * KVD;
        L.qkv_q = dalloc<int8_t>(d, rows * dim);
        L.qkv_s = dalloc<float>(d, rows * dim / G);
        HIPCHK(hipMemcpy(L.qkv_q, w->wq[l].q, P * dim, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(L.qkv_q + P * dim, w->wk[l].q, KVD * dim, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(L.qkv_q + (P + KVD) * dim, w->wv[l].q, KVD * dim, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(L.qkv_s, w->wq[l].s, P * dim / G * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(L.qkv_s + P * dim / G, w->wk[l].s, KVD * dim / G * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(L.qkv_s + (P + KVD) * dim / G, w->wv[l].s, KVD * dim / G * 4, hipMemcpyHostToDevice));
        L.wo_q = upload<int8_t>(d, w->wo[l].q, dim * P);
        L.wo_s = upload<float>(d, w->wo[l].s, dim * P / G);
        L.dn_q = upload<int8_t>(d, w->w2[l].q, dim * hid);
        L.dn_s = upload<float>(d, w->w2[l].s, dim * hid / G);
        L.gu_q = dalloc<int8_t>(d, 2 * hid * dim);
        L.gu_s = dalloc<float>(d, 2 * hid * dim / G);
        for (int which = 0; which < 2; which++) {
            const Q8Tensor* src = which == 0 ? &w->w1[l] : &w->w3[l];
            HIPCHK(hipMemcpy(stage_q, src->q, hid * dim, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(stage_s, src->s, hid * dim / G * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_interleave_rows, dim3(2048), dim3(256), 0, d->st, (char*)L.gu_q, stage_q,
                               dim, hid, which);
            hipLaunchKernelGGL(k_interleave_rows, dim3(256), dim3(256), 0, d->st, (char*)L.gu_s, stage_s,
                               dim / G * 4, hid, which);
            HIPCHK(hipStreamSynchronize(d->st));
        }
        L.att_nw = upload<float>(d, w->att_rms_norm + (size_t)l * dim, dim);
        L.ffn_nw = upload<float>(d, w->ffn_rms_norm + (size_t)l * dim, dim);
        L.qnw = upload<float>(d, w->q_rms_norm + (size_t)l * hd, hd);
        L.knw = upload<float>(d, w->k_rms_norm + (size_t)l * hd, hd);
        const size_t cache = (size_t)d->KV * d->seq_pad * hd;
        d->cache_floats = cache;
        L.kc = dalloc<float>(d, cache * d->n_streams);
        L.vc = dalloc<float>(d, cache * d->n_streams);
        HIPCHK(hipMemsetAsync(L.kc, 0, cache * d->n_streams * 4, d->st));
        HIPCHK(hipMemsetAsync(L.vc, 0, cache * d->n_streams * 4, d->st));
    }
    HIPCHK(hipStreamSynchronize(d->st));
    HIPCHK(hipFree(stage_q));
    HIPCHK(hipFree(stage_s));
    const bool tied = m->params.shared_classifier != 0;
    if (d->has_embed || (tied && d->has_cls)) {
        d->emb_q = upload<int8_t>(d, w->qe->q, V * dim);
        d->emb_s = upload<float>(d, w->qe->s, V * dim / G);
    }
    if (d->has_cls) {
        if (tied) {
            d->cls_q = d->emb_q;
            d->cls_s = d->emb_s;
        } else {
            d->cls_q = upload<int8_t>(d, w->cls->q, V * dim);
            d->cls_s = upload<float>(d, w->cls->s, V * dim / G);
        }
        d->out_nw = upload<float>(d, w->out_rms_norm, dim);
    }
}

void pipeline_split(const ModelParams* p, int rank, int world, int* first, int* count) {
    q3_pipeline_layers(p, rank, world, first, count);
}

void destroy_dev(Dev* d);

struct AttachOpts {
    bool fp16 = false;
    int stage_rank = 0, stage_world = 0;     // loopback self-test: this Dev is stage rank of world
};

Dev* attach(Model* m, const AttachOpts& opt = AttachOpts()) {
    // creation is serialised (two threads calling forward() on a new Model must not both upload it); the
    // registry lock itself is only held for look-ups
    static std::mutex create_mu;
    std::lock_guard<std::mutex> create_lk(create_mu);
    {
        Dev* stale = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            auto it = g_reg.find(m);
            if (it != g_reg.end()) {
                if (it->second->id == model_id(m)) return it->second;
                stale = it->second;           // same address, different Model: the old one was freed behind our back
                g_reg.erase(it);
            }
        }
        if (stale) destroy_dev(stale);
    }
    die_if_no_gpu();
    const ModelParams* p = &m->params;
    if (p->block_size != Q3_GROUP) Q3_DIE("group size %d is not 64", p->block_size);
    if (p->head_dim != 64 && p->head_dim != 128) Q3_DIE("head_dim %d not supported (64 or 128)", p->head_dim);
    if (p->dim % 64 || p->hidden_dim % 64) Q3_DIE("dim/hidden_dim must be multiples of 64");
    Dev* d = new Dev();
    d->m = m;
    d->id = model_id(m);
    d->logits_host = m->state.logits;
    d->device = pick_device();
    HIPCHK(hipSetDevice(d->device));
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, d->device));
        q3k::set_cu_count(prop.multiProcessorCount);
    }
    HIPCHK(hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking));
    d->dim = p->dim; d->hid = p->hidden_dim; d->L = p->n_layers; d->H = p->n_heads;
    d->KV = p->n_kv_heads; d->hd = p->head_dim; d->V = p->vocab_size; d->seq = p->seq_len;
    d->P = d->H * d->hd; d->KVD = d->KV * d->hd;
    d->seq_pad = (d->seq + Q3_ATT_CHUNK - 1) / Q3_ATT_CHUNK * Q3_ATT_CHUNK;
    d->l0 = 0; d->l1 = d->L;
    if (opt.stage_world > 0) {
        d->rank = opt.stage_rank; d->world = opt.stage_world; d->loopback = true;
    } else if (g_pipe.on) {
        d->rank = g_pipe.rank; d->world = g_pipe.world;
    }
    if (d->world > 1) {
        int first, count;
        pipeline_split(p, d->rank, d->world, &first, &count);
        d->l0 = first; d->l1 = first + count;
        d->has_embed = d->rank == 0;
        d->has_cls = d->rank == d->world - 1;
        d->n_streams = d->world;
    }
    const char* eg = getenv("Q3_GRAPH");
    d->use_graph = !(eg && eg[0] == '0');

    upload_weights(d);
    {
        const char* ef = getenv("Q3_FP16");
        d->fp16 = opt.fp16 || (ef && ef[0] == '1');
        if (d->fp16) {
            auto conv = [&](const int8_t* q, const float* sc, size_t elems) {
                void* h = dalloc<uint16_t>(d, elems);
                q3k::to_half(q, sc, elems, h, d->st);
                return h;
            };
            const size_t QKV = (size_t)d->P + 2 * d->KVD;
            for (int l = d->l0; l < d->l1; l++) {
                LayerDev& L = d->layers[l];
                L.qkv_h = conv(L.qkv_q, L.qkv_s, QKV * d->dim);
                L.wo_h = conv(L.wo_q, L.wo_s, (size_t)d->dim * d->P);
                L.gu_h = conv(L.gu_q, L.gu_s, (size_t)2 * d->hid * d->dim);
                L.dn_h = conv(L.dn_q, L.dn_s, (size_t)d->dim * d->hid);
            }
            if (d->has_embed) d->emb_h = conv(d->emb_q, d->emb_s, (size_t)d->V * d->dim);
            if (d->has_cls) d->cls_h = (d->cls_q == d->emb_q && d->emb_h) ? d->emb_h : conv(d->cls_q, d->cls_s, (size_t)d->V * d->dim);
        }
    }
    build_rope(d);
    d->x = dalloc<float>(d, d->dim + 1);      // + the token slot of pipeline messages
    d->xout = dalloc<float>(d, d->dim + 1);
    HIPCHK(hipMemsetAsync(d->xout, 0, ((size_t)d->dim + 1) * 4, d->st));
    d->qkv = dalloc<float>(d, d->P + 2 * d->KVD);
    d->h = dalloc<float>(d, d->hid);
    d->logits = dalloc<float>(d, d->V);
    d->att_q = dalloc<int8_t>(d, d->P);
    d->att_s = dalloc<float>(d, d->P / 64);
    d->att_f = dalloc<float>(d, d->P);
    d->cs_cur = dalloc<float>(d, d->hd);
    d->max_chunks = (d->seq + Q3_ATT_CHUNK - 1) / Q3_ATT_CHUNK;
    // one workgroup per (kv head, chunk) up to 4096 positions = 512 workgroups, two per CU; beyond, the
    // workgroups walk several chunks each instead of queueing a few stragglers behind a full first round
    d->chunk_slots = d->max_chunks < 64 ? d->max_chunks : 64;
    d->part = dalloc<float>(d, (size_t)d->H * d->max_chunks * (d->hd + 2));
    d->tap_dev = dalloc<float>(d, (size_t)d->L * d->dim);
    d->ctl = dalloc<q3k::Ctl>(d, 1);
    d->amax = dalloc<int>(d, 1);
    d->amax_scratch = dalloc<float>(d, 256);
    d->tickets = dalloc<unsigned>(d, d->KV);
    HIPCHK(hipMemsetAsync(d->tickets, 0, (size_t)d->KV * sizeof(unsigned), d->st));
    if (d->loopback) {
        for (int i = 0; i < 2; i++) d->outbox[i] = dalloc<float>(d, d->dim + 1);
    }
    d->nshapes = q3k::step_shapes(d->seq);
    d->gexec.assign((size_t)d->nshapes, nullptr);
    d->pgexec.assign((size_t)d->n_streams * d->nshapes, nullptr);
    d->epoch = dalloc<unsigned>(d, 4);
    HIPCHK(hipMemsetAsync(d->epoch, 0, 4 * sizeof(unsigned), d->st));
    d->att_g = dalloc<unsigned long long>(d, (size_t)d->P / 4 + d->P / 64);
    HIPCHK(hipMemsetAsync(d->att_g, 0, ((size_t)d->P / 4 + d->P / 64) * 8, d->st));     // tag 0 = never a step's tag
    d->part_g = dalloc<unsigned long long>(d, (size_t)d->H * d->max_chunks * (d->hd + 2));
    HIPCHK(hipMemsetAsync(d->part_g, 0, (size_t)d->H * d->max_chunks * (d->hd + 2) * 8, d->st));
    HIPCHK(hipHostMalloc((void**)&d->err_host, 4 * sizeof(unsigned), hipHostMallocMapped));
    memset(d->err_host, 0, 4 * sizeof(unsigned));
    {
        const char* ef = getenv("Q3_FUSE");
        d->fuse = !(ef && ef[0] == '0');
        const char* ed = getenv("Q3_WO_DELAY");
        if (ed) d->pf_delay = atoi(ed);
    }
    if (getenv("Q3_STAMPS")) { d->stamps = dalloc<unsigned long long>(d, 16384); HIPCHK(hipMemset(d->stamps, 0, 16384 * 8)); }
    HIPCHK(hipHostMalloc((void**)&d->ctl_host, sizeof(q3k::Ctl), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&d->amax_host, sizeof(int), hipHostMallocDefault));
    if (d->logits_host) {
        hipError_t e = hipHostRegister(d->logits_host, (size_t)d->V * sizeof(float), hipHostRegisterDefault);
        d->logits_pinned = (e == hipSuccess);
        if (!d->logits_pinned) (void)hipGetLastError();
    }
    d->tap_host.assign((size_t)d->L * d->dim, 0.0f);
    HIPCHK(hipStreamSynchronize(d->st));

    std::lock_guard<std::mutex> lk(g_mu);
    g_reg[m] = d;
    if (!g_atexit) {
        g_atexit = true;
        atexit([]() {
            std::vector<Model*> ms;
            {
                std::lock_guard<std::mutex> lk2(g_mu);
                for (auto& kv : g_reg) ms.push_back(kv.first);
            }
            for (Model* mm : ms) q3_device_detach(mm);
        });
    }
    return d;
}

// Tear a device context down.  Never dereferences the Model: by the time this runs (atexit, or a stale
// registry entry) the reference's model_free() may have released it.
void destroy_dev(Dev* d) {
    (void)hipSetDevice(d->device);
    (void)hipStreamSynchronize(d->st);
    for (auto& ex : d->gexec) {
        if (ex) (void)hipGraphExecDestroy(ex);
    }
    for (auto& ex : d->pgexec) {
        if (ex) (void)hipGraphExecDestroy(ex);
    }
    for (auto& pr : d->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (d->logits_pinned) {
        // (if the Model was freed first, the range is gone already and the runtime says so: ignored)
        if (hipHostUnregister(d->logits_host) != hipSuccess) (void)hipGetLastError();
    }
    for (void* p : d->allocs) (void)hipFree(p);
    (void)hipHostFree(d->ctl_host);
    (void)hipHostFree(d->amax_host);
    if (d->err_host) (void)hipHostFree(d->err_host);
    if (d->samp_tok_host) (void)hipHostFree(d->samp_tok_host);
    (void)hipStreamDestroy(d->st);
    delete d;
}

// the device context of THIS Model, or null (an entry left behind by a Model that was freed without
// q3_device_detach() and whose address has been reused does not count)
Dev* lookup(Model* m) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_reg.find(m);
    if (it == g_reg.end() || !(it->second->id == model_id(m))) return nullptr;
    return it->second;
}

// ---- profiling helpers ----------------------------------------------------

int prof_slot(Dev* d, const char* name, double bytes) {
    for (size_t i = 0; i < d->prof_slots.size(); i++) {
        if (!strcmp(d->prof_slots[i].name, name)) return (int)i;
    }
    d->prof_slots.push_back({name, bytes, 0, 0.0, 0.0, 0});
    return (int)d->prof_slots.size() - 1;
}

struct Timed {
    Dev* d;
    bool on;
    size_t idx;
    Timed(Dev* dev, const char* name, double bytes) : d(dev), on(dev->prof), idx(0) {
        if (!on) return;
        if (d->ev_used == d->ev_pool.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            d->ev_pool.push_back({a, b});
            d->ev_slot.push_back(0);
        }
        idx = d->ev_used++;
        d->ev_slot[idx] = prof_slot(d, name, bytes);
        HIPCHK(hipEventRecord(d->ev_pool[idx].first, d->st));
    }
    // device-clock slot of this bracket for kernels that support it (null when out of slots)
    unsigned long long* clk() const {
        return (on && d->clk_dev && idx < (size_t)CLK_MAX) ? d->clk_dev + 2 * idx : nullptr;
    }
    ~Timed() {
        if (on) HIPCHK(hipEventRecord(d->ev_pool[idx].second, d->st));
    }
};

void prof_begin(Dev* d) {      // before the launches of a profiled step
    if (!d->prof) return;
    if (!d->clk_dev) {
        d->clk_dev = dalloc<unsigned long long>(d, 2 * CLK_MAX);
        d->clk_host.resize(2 * CLK_MAX);
    }
    for (int i = 0; i < CLK_MAX; i++) {
        d->clk_host[2 * i] = ~0ull;
        d->clk_host[2 * i + 1] = 0ull;
    }
    HIPCHK(hipMemcpyAsync(d->clk_dev, d->clk_host.data(), d->clk_host.size() * 8, hipMemcpyHostToDevice, d->st));
}

void prof_collect(Dev* d) {
    if (!d->prof) return;
    HIPCHK(hipStreamSynchronize(d->st));
    if (d->clk_dev) HIPCHK(hipMemcpy(d->clk_host.data(), d->clk_dev, d->clk_host.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < d->ev_used; i++) {
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, d->ev_pool[i].first, d->ev_pool[i].second));
        ProfSlot& s = d->prof_slots[d->ev_slot[i]];
        s.launches++;
        s.ms += ms;
        if (d->clk_dev && i < (size_t)CLK_MAX && d->clk_host[2 * i + 1] > d->clk_host[2 * i]
            && d->clk_host[2 * i] != ~0ull) {
            s.dev_ms += (double)(d->clk_host[2 * i + 1] - d->clk_host[2 * i]) * 1e-5;   // 100 MHz ticks
            s.dev_launches++;
        }
    }
    d->ev_used = 0;
}

// ---- the decode step --------------------------------------------------------

q3k::Attn attn_args(Dev* d, int l, int stream = 0) {
    const LayerDev& L = d->layers[l];
    q3k::Attn a;
    memset(&a, 0, sizeof(a));      // single position: nz and every z stride 0
    a.ctl = d->ctl; a.qkv = d->qkv; a.qnw = L.qnw; a.knw = L.knw; a.cs = d->cs_cur;
    a.kc = L.kc + (size_t)stream * d->cache_floats; a.vc = L.vc + (size_t)stream * d->cache_floats; a.part = d->part;
    a.tickets = d->tickets; a.oq = d->att_q; a.os = d->att_s;
    a.of = nullptr; a.qdbg = nullptr; a.prepared = 0; a.stamps = d->stamps;
    a.n_heads = d->H; a.n_kv = d->KV; a.hd = d->hd; a.seq_len = d->seq_pad; a.max_chunks = d->max_chunks;
    { static const int vh = getenv("Q3_V_HOLD") ? atoi(getenv("Q3_V_HOLD")) : 150; a.v_hold = vh; }
    return a;
}

// fp16 contrast path: the same five stages on binary16 weights and fp32 activations
void enqueue_layer_f16(Dev* d, int l, q3k::AttMode mode, int stream) {
    const LayerDev& L = d->layers[l];
    q3k::gemv_f16(L.qkv_h, d->dim, d->P + 2 * d->KVD, d->x, L.att_nw, d->qkv, q3k::EPI_STORE, d->st);
    q3k::Attn a = attn_args(d, l, stream);
    a.of = d->att_f;                     // fp32 head outputs (the codes are written too and ignored)
    q3k::attn(a, d->chunk_slots, mode, d->st);
    q3k::gemv_f16(L.wo_h, d->P, d->dim, d->att_f, nullptr, d->x, q3k::EPI_RESID, d->st);
    q3k::gemv_f16(L.gu_h, d->dim, 2 * d->hid, d->x, L.ffn_nw, d->h, q3k::EPI_SWIGLU, d->st);
    q3k::gemv_f16(L.dn_h, d->hid, d->dim, d->h, nullptr, d->x, q3k::EPI_RESID, d->st);
}

// The Wo GEMV of layer l as the consumer half of k_attn_wo (fewer than Q3_ATT_LONG cached positions)
q3k::WoView wo_view(Dev* d, int l) {
    const LayerDev& L = d->layers[l];
    q3k::WoView w;
    memset(&w, 0, sizeof(w));
    w.W = L.wo_q; w.S = L.wo_s; w.n = d->P; w.d = d->dim;
    w.x = d->x;
    // tag = (step counter << 8) | layer_tag: the low byte must hold layer + 1 (a model of 255 layers or more takes the
    // separate launches: enqueue_layer)
    w.gran = d->att_g; w.epoch = d->epoch; w.layer_tag = (unsigned)l + 1u;
    w.delay = d->pf_delay;
    { static const unsigned long long wt = getenv("Q3_WAIT_TICKS") ? strtoull(getenv("Q3_WAIT_TICKS"), nullptr, 10) : 500000000ull; w.wait_ticks = wt; }
    w.stamps = d->stamps;
    HIPCHK(hipHostGetDevicePointer((void**)&w.err, d->err_host, 0));
    return w;
}

void enqueue_layer(Dev* d, int l, q3k::AttMode mode, int stream = 0, int rows_cap = 64) {
    if (d->fp16) {
        enqueue_layer_f16(d, l, mode, stream);
        return;
    }
    const LayerDev& L = d->layers[l];
    q3k::Gemv g;
    memset(&g, 0, sizeof(g));
    const char* stamp_which = (d->stamps && l == (d->l0 + d->l1) / 2) ? getenv("Q3_STAMP_GEMV") : nullptr;
    auto stamp_for = [&](const char* name) { return (stamp_which && !strcmp(stamp_which, name)) ? d->stamps : nullptr; };
    {   // rmsnorm + quantise + Wq|Wk|Wv  (reference forward.c:254-262)
        g.W = L.qkv_q; g.S = L.qkv_s; g.n = d->dim; g.d = d->P + 2 * d->KVD;
        g.xf = d->x; g.nw = L.att_nw; g.out = d->qkv;
        Timed t(d, "qkv", q3_gemv_bytes(g.d, g.n));
        g.clk = t.clk();
        g.stamps = stamp_for("qkv");
        q3k::gemv(g, q3k::PRO_NORM, q3k::EPI_STORE, d->st);
    }
    {   // head norms + RoPE + cache append + attention + quantise (forward.c:267-291)
        q3k::Attn a = attn_args(d, l, stream);
        bool fused = false;
        if (d->fuse && l < 255) {
            q3k::WoView w = wo_view(d, l);
            if (mode == q3k::ATT_LONG) w.delay = 0;       // the merge is one round trip long: no reason to hold Wo back
            a.og = d->att_g; a.epoch = d->epoch; a.layer_tag = w.layer_tag; a.pg = d->part_g;
            if (q3k::attn_wo_supported(a, w, d->chunk_slots, mode)) {
                // attention AND Wo + residual in one launch (forward.c:267-298)
                Timed t(d, "attn_wo", q3_gemv_bytes(d->dim, d->P));
                q3k::attn(a, d->chunk_slots, mode, d->st, rows_cap, &w);
                fused = true;
            }
        }
        if (!fused) {
            Timed t(d, "attn", 0.0);
            q3k::attn(a, d->chunk_slots, mode, d->st, rows_cap);
        }
        if (fused) goto after_wo;
    }
    {   // Wo + residual (forward.c:292-298)
        g.W = L.wo_q; g.S = L.wo_s; g.n = d->P; g.d = d->dim;
        g.xq = d->att_q; g.xs = d->att_s; g.out = d->x;
        Timed t(d, "wo", q3_gemv_bytes(g.d, g.n));
        g.clk = t.clk();
        g.stamps = stamp_for("wo");
        q3k::gemv(g, q3k::PRO_Q8, q3k::EPI_RESID, d->st);
    }
after_wo:
    {   // rmsnorm + quantise + gate/up + SwiGLU (forward.c:303-321)
        g.W = L.gu_q; g.S = L.gu_s; g.n = d->dim; g.d = 2 * d->hid;
        g.xf = d->x; g.nw = L.ffn_nw; g.out = d->h;
        Timed t(d, "gateup", q3_gemv_bytes(g.d, g.n));
        g.clk = t.clk();
        g.stamps = stamp_for("gateup");
        q3k::gemv(g, q3k::PRO_NORM, q3k::EPI_SWIGLU, d->st);
    }
    {   // quantise + down + residual (forward.c:326-338)
        g.W = L.dn_q; g.S = L.dn_s; g.n = d->hid; g.d = d->dim;
        g.xf = d->h; g.nw = nullptr; g.out = d->x;
        Timed t(d, "down", q3_gemv_bytes(g.d, g.n));
        g.clk = t.clk();
        g.stamps = stamp_for("down");
        q3k::gemv(g, q3k::PRO_F32, q3k::EPI_RESID, d->st);
    }
    if (d->tap) {
        HIPCHK(hipMemcpyAsync(d->tap_dev + (size_t)l * d->dim, d->x, (size_t)d->dim * 4,
                              hipMemcpyDeviceToDevice, d->st));
    }
}

void enqueue_head(Dev* d) {
    if (d->fp16) {
        q3k::gemv_f16(d->cls_h, d->dim, d->V, d->x, d->out_nw, d->logits, q3k::EPI_STORE, d->st);
        return;
    }
    q3k::Gemv g;
    memset(&g, 0, sizeof(g));
    g.W = d->cls_q; g.S = d->cls_s; g.n = d->dim; g.d = d->V;
    g.xf = d->x; g.nw = d->out_nw; g.out = d->logits;
    Timed t(d, "cls", q3_gemv_bytes(g.d, g.n));
    g.clk = t.clk();
    q3k::gemv(g, q3k::PRO_NORM, q3k::EPI_STORE, d->st);   // forward.c:344-348
}

// everything of one step that runs on this device, between the ctl upload and the logits
void enqueue_step(Dev* d, int pos_shape, int stream = 0) {
    // pos_shape: any position of the launch shape the step is captured for (q3k::step_shape)
    const q3k::AttMode mode = q3k::attn_mode(pos_shape);
    const int rows_cap = q3k::step_rows_cap(pos_shape);
    {
        Timed t(d, "begin", 0.0);
        q3k::begin_step(d->ctl, (d->has_embed && !d->fp16) ? d->emb_q : nullptr, d->emb_s, d->dim, d->x, d->rope, d->hd,
                        d->cs_cur, d->st, d->epoch, d->V);
        if (d->fp16 && d->has_embed) q3k::embed_half(d->ctl, d->emb_h, d->dim, d->x, d->st, d->V);
    }
    for (int l = d->l0; l < d->l1; l++) enqueue_layer(d, l, mode, stream, rows_cap);
    if (d->has_cls) enqueue_head(d);
}

void launch_stage(Dev* d, int pos, int stream);

// A consumer of an in-launch hand-off (k_attn_wo / k_merge_wo) that gave up waiting raises err_host[0]; its step's
// result is void.  The Model then stops using the fused launches: true = that just happened (the caller redoes its work).
bool handoff_failed(Dev* d) {
    if (!d->err_host || !*(volatile unsigned*)d->err_host) return false;
    if (!d->fuse) Q3_DIE("an in-launch hand-off timed out although the fused launches are off");
    fprintf(stderr, "[q3hip] an in-launch hand-off (attention -> Wo) timed out on the device: this Model continues with separate launches\n");
    *(volatile unsigned*)d->err_host = 0;
    d->fuse = false;
    d->fallbacks++;
    for (auto& ex : d->gexec) { if (ex) { HIPCHK(hipGraphExecDestroy(ex)); ex = nullptr; } }
    for (auto& ex : d->pgexec) { if (ex) { HIPCHK(hipGraphExecDestroy(ex)); ex = nullptr; } }
    return true;
}
// Every synchronisation that hands results to the caller goes through here: wait, and if a hand-off gave up meanwhile,
// queue everything since the last clean synchronisation again (now unfused) and wait again.
void wait_step(Dev* d);
void sync_checked(Dev* d) {
    wait_step(d);
    // (a stage of a pipeline cannot redo its ticks alone: the other stages have moved on)
    if (d->world > 1 && d->err_host && *(volatile unsigned*)d->err_host)
        Q3_DIE("an in-launch hand-off timed out on one stage of a %d-stage pipeline: restart with Q3_FUSE=0", d->world);
    if (handoff_failed(d)) {
        std::vector<std::function<void()>> jobs;
        jobs.swap(d->redo);
        d->replaying = true;
        for (auto& j : jobs) j();
        d->replaying = false;
        HIPCHK(hipStreamSynchronize(d->st));
        if (handoff_failed(d)) Q3_DIE("hand-off flag raised again after the fallback");
    }
    d->redo.clear();
}
// remember how to queue a piece of work again (no-op while a replay is running: the jobs run as they are)
void remember(Dev* d, std::function<void()> job) {
    if (d->replaying || !d->fuse) return;           // (without fused launches there is no hand-off that could give up)
    d->redo.push_back(std::move(job));
    if (d->redo.size() >= 4096) sync_checked(d);     // a caller that queues without ever synchronising: settle what there is
}

void fetch_logits_async(Dev* d) {
    HIPCHK(hipMemcpyAsync(d->logits_host, d->logits, (size_t)d->V * 4, hipMemcpyDeviceToHost, d->st));
}

hipGraphExec_t build_graph(Dev* d, int pos, bool with_logits) {
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(d->st, hipStreamCaptureModeThreadLocal));
    HIPCHK(hipMemcpyAsync(d->ctl, d->ctl_host, sizeof(q3k::Ctl), hipMemcpyHostToDevice, d->st));
    enqueue_step(d, pos);
    if (with_logits) fetch_logits_async(d);
    HIPCHK(hipStreamEndCapture(d->st, &graph));
    hipGraphExec_t exec = nullptr;
    HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(graph));
    return exec;
}

void check_step_args(Dev* d, int token, int pos) {
    if (pos < 0 || pos >= d->seq) Q3_DIE("position %d outside the context window [0,%d)", pos, d->seq);
    if (token < 0 || token >= d->V) Q3_DIE("token %d outside the vocabulary [0,%d)", token, d->V);
}

// The wait at the end of a token step is on the caller's critical path (the next token depends on these
// logits).  Default: the blocking hipStreamSynchronize (measured on a quiet host: within 0.2 % of polling).
// Q3_SPIN=1 polls hipStreamQuery in a pause loop instead -- for hosts whose interrupt wake-ups are slow; it
// keeps one core busy for the ~1.5 ms of every step.
void wait_step(Dev* d) {
    static const bool spin = getenv("Q3_SPIN") && atoi(getenv("Q3_SPIN")) != 0;
    if (!spin) {
        HIPCHK(hipStreamSynchronize(d->st));
        return;
    }
    for (;;) {
        const hipError_t e = hipStreamQuery(d->st);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIPCHK(e);
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// Queue one step whose {token, pos} travel as immediate arguments of a one-thread kernel (q3_forward_device): the caller
// may queue several of these without a sync, so they must not go through the one pinned host slot a later call would
// overwrite before this step's copy has run.  The step is the graph WITHOUT the ctl upload and without the logits download.
void enqueue_async_step(Dev* d, int token, int pos) {
    HIPCHK(hipSetDevice(d->device));
    prof_begin(d);
    q3k::set_ctl(d->ctl, nullptr, token, pos, d->st);
    launch_stage(d, pos, 0);
    if (d->tap) {
        HIPCHK(hipMemcpyAsync(d->tap_host.data(), d->tap_dev, d->tap_host.size() * 4, hipMemcpyDeviceToHost, d->st));
    }
    prof_collect(d);
}

// Queue one synchronous step (forward()): ctl through the pinned slot (read before the call returns), logits downloaded
// behind it when the host buffer is pinned.
void enqueue_sync_step(Dev* d, int token, int pos) {
    HIPCHK(hipSetDevice(d->device));
    d->ctl_host->token = token;
    d->ctl_host->pos = pos;
    if (d->use_graph && !d->prof && !d->tap && d->logits_pinned) {
        if (!d->gexec[q3k::step_shape(pos)]) {
            // first step of this Model: capture the launch shapes of the first Q3_ATT_LONG + 1024 positions now, so that
            // no token pays for a graph instantiation when the position crosses into the next shape (beyond that: one
            // instantiation per 1024 positions)
            const int mine = q3k::step_shape(pos);
            for (int k = 0; k < d->nshapes; k++) {
                if (d->gexec[k] || !(k == mine || (k <= 5 && q3k::step_shape_pos(k) < d->seq))) continue;
                d->gexec[k] = build_graph(d, k == mine ? pos : q3k::step_shape_pos(k), true);
            }
        }
        HIPCHK(hipGraphLaunch(d->gexec[q3k::step_shape(pos)], d->st));
        return;
    }
    prof_begin(d);
    HIPCHK(hipMemcpyAsync(d->ctl, d->ctl_host, sizeof(q3k::Ctl), hipMemcpyHostToDevice, d->st));
    enqueue_step(d, pos);
    if (d->tap) {
        HIPCHK(hipMemcpyAsync(d->tap_host.data(), d->tap_dev, d->tap_host.size() * 4, hipMemcpyDeviceToHost, d->st));
    }
    if (d->logits_pinned) fetch_logits_async(d);
}

// run one step; logits stay on the device unless `to_host`
void run_step(Dev* d, int token, int pos, bool to_host) {
    check_step_args(d, token, pos);
    if (d->world > 1) Q3_DIE("this Model is one stage of a %d-stage pipeline: use q3_pipeline_run()", d->world);
    if (!to_host) {
        enqueue_async_step(d, token, pos);
        remember(d, [d, token, pos] { enqueue_async_step(d, token, pos); });
        return;
    }
    enqueue_sync_step(d, token, pos);
    remember(d, [d, token, pos] { enqueue_sync_step(d, token, pos); });
    sync_checked(d);
    if (!d->logits_pinned) HIPCHK(hipMemcpy(d->logits_host, d->logits, (size_t)d->V * 4, hipMemcpyDeviceToHost));
    if (!(d->use_graph && !d->prof && !d->tap && d->logits_pinned)) prof_collect(d);
}

// ---- context for the stand-alone ops ---------------------------------------
struct OpsCtx {
    bool ready = false;
    hipStream_t st = nullptr;
};
OpsCtx g_ops;

hipStream_t ops_stream() {
    if (!g_ops.ready) {
        die_if_no_gpu();
        HIPCHK(hipSetDevice(pick_device()));
        HIPCHK(hipStreamCreateWithFlags(&g_ops.st, hipStreamNonBlocking));
        g_ops.ready = true;
    }
    return g_ops.st;
}

// Device buffers of the op hooks.  The reference's own code calls these hooks in its inner loops
// (sample() -> softmax(logits, V) once per token, src/sampler.c:196), so a call must not pay for
// hipMalloc / hipFree: released blocks go to a small cache and the next call of the same size gets
// them back.  (The cache holds at most DBUF_CACHE_MAX blocks; larger one-off requests, e.g. the
// dequantisation of a whole embedding table at load time, are really freed.)
constexpr size_t DBUF_CACHE_MAX = 24;
constexpr size_t DBUF_CACHE_BLOCK_LIMIT = 64u << 20;
std::mutex g_dbuf_mu;
std::vector<std::pair<size_t, void*>> g_dbuf_cache;      // (capacity, pointer)

void* dbuf_take(size_t bytes, size_t* cap) {
    {
        std::lock_guard<std::mutex> lk(g_dbuf_mu);
        size_t best = g_dbuf_cache.size();
        for (size_t i = 0; i < g_dbuf_cache.size(); i++) {
            if (g_dbuf_cache[i].first >= bytes && (best == g_dbuf_cache.size() || g_dbuf_cache[i].first < g_dbuf_cache[best].first)) best = i;
        }
        if (best < g_dbuf_cache.size() && g_dbuf_cache[best].first <= 2 * bytes + 4096) {
            void* p = g_dbuf_cache[best].second;
            *cap = g_dbuf_cache[best].first;
            g_dbuf_cache.erase(g_dbuf_cache.begin() + best);
            return p;
        }
    }
    void* p = nullptr;
    *cap = bytes ? bytes : 16;
    HIPCHK(hipMalloc(&p, *cap));
    return p;
}
void dbuf_give(void* p, size_t cap) {
    if (cap <= DBUF_CACHE_BLOCK_LIMIT) {
        std::lock_guard<std::mutex> lk(g_dbuf_mu);
        if (g_dbuf_cache.size() < DBUF_CACHE_MAX) {
            g_dbuf_cache.push_back({cap, p});
            return;
        }
    }
    (void)hipFree(p);
}

struct DBuf {   // RAII device buffer for the op hooks
    void* p = nullptr;
    size_t bytes = 0, cap = 0;
    explicit DBuf(size_t b) : bytes(b) { p = dbuf_take(b, &cap); }
    DBuf(const void* host, size_t b) : bytes(b) {
        p = dbuf_take(b, &cap);
        if (b) HIPCHK(hipMemcpy(p, host, b, hipMemcpyHostToDevice));
    }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { dbuf_give(p, cap); }
    void to_host(void* host, hipStream_t st) {
        HIPCHK(hipStreamSynchronize(st));
        if (bytes) HIPCHK(hipMemcpy(host, p, bytes, hipMemcpyDeviceToHost));
    }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

void rope_cs_host(int hd, int pos, std::vector<float>& cs) {
    const int half = hd / 2;
    cs.resize((size_t)hd);
    for (int i = 0; i < half; i++) {
        const float angle = pos * powf(1e6f, -(float)i / half);
        cs[2 * i] = cosf(angle);
        cs[2 * i + 1] = sinf(angle);
    }
}

}  // namespace

namespace { void ensure_prefill(Dev* d); }
#ifndef Q3_PF_CHUNK
#define Q3_PF_CHUNK 64
#endif

// ============================================================ C ABI =========
extern "C" {

const char* q3_version(void) { return "q3hip 0.1 (gfx950)"; }

// diagnostic: copy the attention kernel's time stamps (Q3_STAMPS=1 + -DQ3_ATTN_STAMPS build)
int q3_debug_stamps(Model* m, unsigned long long* out, int n) {
    Dev* d = lookup(m);
    if (!d || !d->stamps) return 0;
    HIPCHK(hipStreamSynchronize(d->st));
    HIPCHK(hipMemcpy(out, d->stamps, (size_t)(n < 16384 ? n : 16384) * 8, hipMemcpyDeviceToHost));
    return n < 16384 ? n : 16384;
}

// diagnostic: `iters` back-to-back launches of one GEMV class cycling over layers [l_lo, l_hi),
// HIP-event timed as a whole; mean microseconds per launch.  With l_hi == l_lo + 1 the same
// weights are re-read every launch (Infinity-Cache resident), otherwise each launch streams fresh ones.
double q3_debug_gemv_loop(Model* m, const char* which, int l_lo, int l_hi, int iters) {
    Dev* d = lookup(m);
    if (!d || l_lo < d->l0 || l_hi > d->l1 || l_hi <= l_lo || iters <= 0) return -1.0;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int pass = 0; pass < 2; pass++) {      // pass 0 warms up
        HIPCHK(hipEventRecord(e0, d->st));
        for (int i = 0; i < iters; i++) {
            const LayerDev& L = d->layers[l_lo + i % (l_hi - l_lo)];
            q3k::Gemv g;
            memset(&g, 0, sizeof(g));
            if (!strcmp(which, "qkv")) {
                g.W = L.qkv_q; g.S = L.qkv_s; g.n = d->dim; g.d = d->P + 2 * d->KVD;
                g.xf = d->x; g.nw = L.att_nw; g.out = d->qkv;
                q3k::gemv(g, q3k::PRO_NORM, q3k::EPI_STORE, d->st);
            } else if (!strcmp(which, "wo")) {
                g.W = L.wo_q; g.S = L.wo_s; g.n = d->P; g.d = d->dim;
                g.xq = d->att_q; g.xs = d->att_s; g.out = d->qkv;       // scratch output
                q3k::gemv(g, q3k::PRO_Q8, q3k::EPI_STORE, d->st);
            } else if (!strcmp(which, "gateup")) {
                g.W = L.gu_q; g.S = L.gu_s; g.n = d->dim; g.d = 2 * d->hid;
                g.xf = d->x; g.nw = L.ffn_nw; g.out = d->h;
                q3k::gemv(g, q3k::PRO_NORM, q3k::EPI_SWIGLU, d->st);
            } else {
                g.W = L.dn_q; g.S = L.dn_s; g.n = d->hid; g.d = d->dim;
                g.xf = d->h; g.nw = nullptr; g.out = d->qkv;
                q3k::gemv(g, q3k::PRO_F32, q3k::EPI_STORE, d->st);
            }
        }
        HIPCHK(hipEventRecord(e1, d->st));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    }
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    return (double)ms * 1e3 / iters;
}

// diagnostic: the same back-to-back loop through the int8-MFMA GEMM of the prompt path (q3_prefill.hip) on `ntok`
// quantised activation rows (1 = the decode GEMV's work on the matrix cores, bit-identical results): mean us per launch
double q3_debug_gemm_loop(Model* m, const char* which, int ntok, int l_lo, int l_hi, int iters) {
    Dev* d = lookup(m);
    if (!d || d->fp16 || l_lo < d->l0 || l_hi > d->l1 || l_hi <= l_lo || iters <= 0 || ntok < 1 || ntok > Q3_PF_CHUNK) return -1.0;
    HIPCHK(hipSetDevice(d->device));
    ensure_prefill(d);
    HIPCHK(hipMemsetAsync(d->pf_xq, 1, (size_t)Q3_PF_CHUNK * 64, d->st));       // any finite codes / scales will do for timing
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int pass = 0; pass < 2; pass++) {      // pass 0 warms up
        HIPCHK(hipEventRecord(e0, d->st));
        for (int i = 0; i < iters; i++) {
            const LayerDev& L = d->layers[l_lo + i % (l_hi - l_lo)];
            if (!strcmp(which, "qkv")) {
                q3k::gemm_q8(L.qkv_q, L.qkv_s, d->dim, d->P + 2 * d->KVD, d->pf_xq, d->pf_xs, ntok, d->pf_qkv, d->P + 2 * d->KVD, q3k::EPI_STORE, d->st);
            } else if (!strcmp(which, "wo")) {
                q3k::gemm_q8(L.wo_q, L.wo_s, d->P, d->dim, d->pf_xq, d->pf_xs, ntok, d->pf_qkv, d->dim, q3k::EPI_STORE, d->st);
            } else if (!strcmp(which, "gateup")) {
                q3k::gemm_q8(L.gu_q, L.gu_s, d->dim, 2 * d->hid, d->pf_xq, d->pf_xs, ntok, d->pf_h, d->hid, q3k::EPI_SWIGLU, d->st);
            } else {
                q3k::gemm_q8(L.dn_q, L.dn_s, d->hid, d->dim, d->pf_xq, d->pf_xs, ntok, d->pf_qkv, d->dim, q3k::EPI_STORE, d->st);
            }
        }
        HIPCHK(hipEventRecord(e1, d->st));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    }
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    return (double)ms * 1e3 / iters;
}

int q3_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int q3_device_attach(Model* m) {
    if (!m) return -1;
    attach(m);
    return 0;
}

/* q3_device_attach() for the fp16 contrast path (BASELINE config 5): every Q8_0 matrix is dequantised and
 * rounded to binary16 on the device, forward() then runs on those with fp32 activations.  Must be the
 * first call that touches the device for this Model.  (Q3_FP16=1 in the environment does the same.) */
int q3_device_attach_fp16(Model* m) {
    if (!m) return -1;
    if (lookup(m)) Q3_DIE("q3_device_attach_fp16: this Model already has device state");
    AttachOpts o;
    o.fp16 = true;
    attach(m, o);
    return 0;
}

void q3_device_detach(Model* m) {
    Dev* d = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_reg.find(m);
        if (it == g_reg.end()) return;
        d = it->second;
        g_reg.erase(it);
    }
    destroy_dev(d);
}

/* how many times this Model fell back from the fused launches to separate ones because an in-launch hand-off timed out */
int q3_handoff_fallbacks(Model* m) {
    Dev* d = lookup(m);
    return d ? d->fallbacks : 0;
}

void q3_device_sync(Model* m) {
    Dev* d = lookup(m);
    if (d) sync_checked(d);
}

float* forward(Model* m, int token, int pos) {
    if (!m) Q3_DIE("forward: NULL model");
    Dev* d = attach(m);
    run_step(d, token, pos, true);
    return m->state.logits;
}

void q3_forward_device(Model* m, int token, int pos) {
    Dev* d = attach(m);
    run_step(d, token, pos, false);
}

void q3_logits_fetch(Model* m) {
    Dev* d = attach(m);
    sync_checked(d);
    HIPCHK(hipMemcpy(m->state.logits, d->logits, (size_t)d->V * 4, hipMemcpyDeviceToHost));
}

int q3_device_argmax(Model* m) {
    Dev* d = attach(m);
    auto pick = [d] {
        q3k::argmax(d->logits, d->V, d->amax_scratch, d->amax, nullptr, d->st);
        HIPCHK(hipMemcpyAsync(d->amax_host, d->amax, sizeof(int), hipMemcpyDeviceToHost, d->st));
    };
    pick();
    remember(d, pick);
    sync_checked(d);
    return *d->amax_host;
}

}  // extern "C"

namespace {

// ---- on-device token loop and layer pipeline ---------------------------------------
//
// N ranks (one process per GPU) each own a contiguous block of layers; N independent
// token streams travel around the ring so that every stage is busy on every tick:
// stream s, token k sits on rank r at tick t = s + k*N + r.  Hand-offs are RCCL
// point-to-point over xGMI: the fp32 residual x[dim] from rank r to r+1, and the chosen
// token id from the last rank back to rank 0.  With N = 1 the same loop is the
// on-device greedy decoder (no host round trip per token).

void ensure_token_log(Dev* d, int per_stream) {
    if (per_stream <= d->ptokens_cap) return;
    if (d->ptokens) {      // grow: the old log is released, not abandoned
        HIPCHK(hipStreamSynchronize(d->st));
        for (size_t i = 0; i < d->allocs.size(); i++) {
            if (d->allocs[i] == (void*)d->ptokens) { d->allocs.erase(d->allocs.begin() + i); break; }
        }
        HIPCHK(hipFree(d->ptokens));
    }
    d->ptokens = dalloc<int>(d, (size_t)per_stream * d->n_streams);
    d->ptokens_cap = per_stream;
}

__global__ void k_log_token(const int* tok, int* log_slot) { *log_slot = *tok; }

void launch_stage(Dev* d, int pos, int stream) {
    const bool pinned_ok = true;
    if (d->use_graph && !d->prof && !d->tap && pinned_ok) {
        if (!d->pgexec[(size_t)stream * d->nshapes + q3k::step_shape(pos)]) {
            const int mine = q3k::step_shape(pos);                    // the early shapes at once: see run_step
            for (int k = 0; k < d->nshapes; k++) {
                hipGraphExec_t& exk = d->pgexec[(size_t)stream * d->nshapes + k];
                if (exk || !(k == mine || (k <= 5 && q3k::step_shape_pos(k) < d->seq))) continue;
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(d->st, hipStreamCaptureModeThreadLocal));
                enqueue_step(d, k == mine ? pos : q3k::step_shape_pos(k), stream);
                HIPCHK(hipStreamEndCapture(d->st, &graph));
                HIPCHK(hipGraphInstantiate(&exk, graph, nullptr, nullptr, 0));
                HIPCHK(hipGraphDestroy(graph));
            }
        }
        hipGraphExec_t& ex = d->pgexec[(size_t)stream * d->nshapes + q3k::step_shape(pos)];
        HIPCHK(hipGraphLaunch(ex, d->st));
    } else {
        enqueue_step(d, pos, stream);
    }
}

#define NCCLCHK(expr)                                                                   \
    do {                                                                                \
        ncclResult_t r_ = (expr);                                                       \
        if (r_ != ncclSuccess) Q3_DIE("%s failed: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

// Ring exchange at the end of a tick: EVERY rank sends one message (x[dim] + the token
// slot) to its successor and receives one from its predecessor, inside one RCCL group,
// so the hand-off is deadlock-free whatever the transport's buffering is.  Ranks that
// have nothing to say during pipeline fill/drain send a stale buffer the receiver ignores.
// Loopback self-test: the same exchange through device mailboxes indexed by tick parity.
void ring_exchange(Dev* d, int tick) {
    const size_t n = (size_t)d->dim + 1;
    if (d->loopback) {
        HIPCHK(hipMemcpyAsync(d->outbox[tick & 1], d->xout, n * 4, hipMemcpyDeviceToDevice, d->st));
        return;
    }
    const int next = (d->rank + 1) % d->world, prev = (d->rank + d->world - 1) % d->world;
    NCCLCHK(ncclGroupStart());
    NCCLCHK(ncclSend(d->xout, n, ncclFloat32, next, g_pipe.comm, d->st));
    NCCLCHK(ncclRecv(d->x, n, ncclFloat32, prev, g_pipe.comm, d->st));
    NCCLCHK(ncclGroupEnd());
}
void loopback_deliver(Dev* d, int tick) {   // what the recv half of the exchange does
    HIPCHK(hipMemcpyAsync(d->x, d->loop_prev->outbox[tick & 1], ((size_t)d->dim + 1) * 4, hipMemcpyDeviceToDevice, d->st));
}

// The compute half of one tick of one stage (q3_pipeline_schedule says which stream /
// token it works on).  The incoming message sits in d->x: the residual for ranks > 0,
// the token chosen by the last rank (slot x[dim]) for rank 0.
void pipeline_tick(Dev* d, int first_token, int pos0, int s, int k) {
    const int N = d->world, r = d->rank;
    const int pos = pos0 + k;
    const bool last = r == N - 1;
    int* tok_slot_in = reinterpret_cast<int*>(d->x + d->dim);
    int* tok_slot_out = reinterpret_cast<int*>(d->xout + d->dim);
    HIPCHK(hipSetDevice(d->device));
    q3k::set_ctl(d->ctl, (r == 0 && k > 0) ? tok_slot_in : nullptr, r == 0 ? first_token : 0, pos, d->st);
    launch_stage(d, pos, s);
    if (!last) {
        HIPCHK(hipMemcpyAsync(d->xout, d->x, (size_t)d->dim * 4, hipMemcpyDeviceToDevice, d->st));
    } else {
        // greedy pick on the device; with one stage it feeds the next step directly
        int* dst = (N == 1 && !g_pipe.self) ? tok_slot_in : tok_slot_out;
        if (d->samp_on) q3k::sample(d->logits, d->V, d->samp_t, d->samp_p, 0.0f, d->seed_dev, d->sb, dst, nullptr, d->st);
        else q3k::argmax(d->logits, d->V, d->amax_scratch, dst, nullptr, d->st);
        hipLaunchKernelGGL(k_log_token, dim3(1), dim3(1), 0, d->st, dst, d->ptokens + (size_t)s * d->ptokens_cap + k);
    }
}

void pipeline_check(Dev* d, int first_token, int pos0, int nsteps) {
    if (pos0 < 0 || nsteps < 0 || pos0 + nsteps > d->seq) Q3_DIE("pipeline_run: positions [%d,%d) outside the window %d", pos0, pos0 + nsteps, d->seq);
    if (first_token < 0 || first_token >= d->V) Q3_DIE("pipeline_run: bad first token %d", first_token);
    if (d->rank == d->world - 1) ensure_token_log(d, nsteps);
}

// Runs `nsteps` tokens of the first `streams` streams (0 or >= world: all of them), starting from
// `first_token` at `pos0`.  The tick schedule is the same whatever `streams` is -- a stream that does not
// run leaves its ticks idle -- so streams = 1 is ONE token stream travelling through the stages, each
// stage busy on one tick in `world` (the single-stream figure).
// Returns the number of ticks on which this rank computed.
int pipeline_run(Dev* d, int first_token, int pos0, int nsteps, int streams = 0) {
    pipeline_check(d, first_token, pos0, nsteps);
    if (streams <= 0 || streams > d->world) streams = d->world;
    if (d->world == 1) {      // (the on-device token loop: redone whole if a hand-off gives up; the sampler's state travels with it)
        const bool samp = d->samp_on;
        const float st = d->samp_t, sp = d->samp_p;
        remember(d, [d, first_token, pos0, nsteps, streams, samp, st, sp] {
            const bool on = d->samp_on;
            const float t0 = d->samp_t, p0 = d->samp_p;
            d->samp_on = samp; d->samp_t = st; d->samp_p = sp;
            pipeline_run(d, first_token, pos0, nsteps, streams);
            d->samp_on = on; d->samp_t = t0; d->samp_p = p0;
        });
    }
    int ticks = 0;
    const int T = nsteps * d->world + d->world - 1;
    for (int t = 0; t < T; t++) {
        int s = 0, k = 0;
        if (q3_pipeline_schedule(d->rank, d->world, nsteps, t, &s, &k) && s < streams) {
            pipeline_tick(d, first_token, pos0, s, k);
            ticks++;
        }
        if ((d->world > 1 || g_pipe.self) && t < T - 1) ring_exchange(d, t);
    }
    return ticks;
}

}  // namespace

extern "C" {

int q3_pipeline_run(Model* m, int first_token, int pos0, int nsteps) {
    Dev* d = attach(m);
    return pipeline_run(d, first_token, pos0, nsteps);
}

int q3_pipeline_run_streams(Model* m, int first_token, int pos0, int nsteps, int streams) {
    Dev* d = attach(m);
    return pipeline_run(d, first_token, pos0, nsteps, streams);
}

// Tokens chosen by stream `stream` in the last q3_pipeline_run (valid on the last rank).
int q3_pipeline_tokens(Model* m, int stream, int* out, int n) {
    Dev* d = attach(m);
    if (!d->ptokens || stream < 0 || stream >= d->n_streams) return 0;
    if (n > d->ptokens_cap) n = d->ptokens_cap;
    sync_checked(d);
    HIPCHK(hipMemcpy(out, d->ptokens + (size_t)stream * d->ptokens_cap, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return n;
}

int q3_generate_greedy(Model* m, int token, int pos, int n, int* out_tokens) {
    Dev* d = attach(m);
    if (d->world > 1) Q3_DIE("q3_generate_greedy: use q3_pipeline_run on a pipeline");
    if (pos + n > d->seq) n = d->seq - pos;
    if (n <= 0) return 0;
    pipeline_run(d, token, pos, n);
    if (out_tokens) q3_pipeline_tokens(m, 0, out_tokens, n);
    q3_logits_fetch(m);
    return n;
}

// ---- batched prompt ingestion ------------------------------------------
#ifndef Q3_PF_CHUNK
#define Q3_PF_CHUNK 64
#endif
namespace {
void ensure_prefill(Dev* d) {
    if (d->pf_ready) return;
    const int B = Q3_PF_CHUNK;
    int wide = d->dim > d->hid ? d->dim : d->hid;
    if (d->P > wide) wide = d->P;
    d->pf_tokens = dalloc<int>(d, B);
    d->pf_ctl = dalloc<q3k::Ctl>(d, B);
    d->pf_cs = dalloc<float>(d, (size_t)B * d->hd);
    d->pf_x = dalloc<float>(d, (size_t)B * d->dim);
    d->pf_qkv = dalloc<float>(d, (size_t)B * (d->P + 2 * d->KVD));
    d->pf_h = dalloc<float>(d, (size_t)B * d->hid);
    d->pf_aq = dalloc<int8_t>(d, (size_t)B * d->P);
    d->pf_as = dalloc<float>(d, (size_t)B * d->P / 64);
    d->pf_xq = dalloc<int8_t>(d, (size_t)B * wide);
    d->pf_xs = dalloc<float>(d, (size_t)B * wide / 64);
    d->pf_part = dalloc<float>(d, (size_t)B * d->H * d->max_chunks * (d->hd + 2));
    d->pf_tickets = dalloc<unsigned>(d, (size_t)B * d->KV);
    HIPCHK(hipMemsetAsync(d->pf_tickets, 0, (size_t)B * d->KV * sizeof(unsigned), d->st));
    if (d->fp16) {
        d->pf_xh = dalloc<uint16_t>(d, (size_t)B * wide);
        d->pf_attf = dalloc<float>(d, (size_t)B * d->P);
    }
    d->pf_ready = true;
}

// one chunk of bc <= Q3_PF_CHUNK consecutive positions through every layer of this device
void prefill_chunk(Dev* d, const int* tokens, int bc, int pos0) {
    const int QKV = d->P + 2 * d->KVD;
    HIPCHK(hipMemcpyAsync(d->pf_tokens, tokens, (size_t)bc * sizeof(int), hipMemcpyHostToDevice, d->st));
    q3k::prefill_begin(d->pf_tokens, bc, pos0, d->emb_q, d->emb_s, d->dim, d->pf_x, d->dim, d->rope, d->hd, d->pf_cs,
                       d->pf_ctl, d->st);
    for (int l = d->l0; l < d->l1; l++) {
        const LayerDev& L = d->layers[l];
        // rmsnorm + quantise of the 16 residual rows, then Wq|Wk|Wv on the matrix cores (forward.c:254-262)
        q3k::rows_quantize(d->pf_x, d->dim, L.att_nw, d->dim, bc, d->pf_xq, d->pf_xs, d->st);
        q3k::gemm_q8(L.qkv_q, L.qkv_s, d->dim, QKV, d->pf_xq, d->pf_xs, bc, d->pf_qkv, QKV, q3k::EPI_STORE, d->st);
        // k/v of all the positions into the cache first (a later position attends to an earlier one's),
        // then the decode attention kernel with one grid layer per position -- one launch per run of
        // positions that share a launch shape
        {
            q3k::Attn a = attn_args(d, l, 0);
            a.ctl = d->pf_ctl; a.qkv = d->pf_qkv; a.cs = d->pf_cs; a.oq = d->pf_aq; a.os = d->pf_as;
            a.part = d->pf_part; a.tickets = d->pf_tickets;
            a.zs_qkv = QKV; a.zs_cs = d->hd; a.zs_oq = d->P; a.zs_os = d->P / 64; a.zs_tickets = d->KV;
            a.zs_part = (size_t)d->H * d->max_chunks * (d->hd + 2);
            q3k::kv_append(a, bc, d->st);
            for (int t0 = 0; t0 < bc;) {
                const q3k::AttMode mode = q3k::attn_mode(pos0 + t0);
                int t1 = t0 + 1;
                while (t1 < bc && q3k::attn_mode(pos0 + t1) == mode) t1++;
                q3k::Attn b = a;
                b.ctl += t0; b.qkv += (size_t)t0 * a.zs_qkv; b.cs += (size_t)t0 * a.zs_cs; b.oq += (size_t)t0 * a.zs_oq;
                b.os += (size_t)t0 * a.zs_os; b.part += (size_t)t0 * a.zs_part; b.tickets += (size_t)t0 * a.zs_tickets;
                b.nz = t1 - t0;
                // the positions are known here (no graph to keep in shape): only the chunk slots the last one needs
                const int need = (pos0 + t1 - 1) / Q3_ATT_CHUNK + 1;
                q3k::attn(b, need < d->chunk_slots ? need : d->chunk_slots, mode, d->st);
                t0 = t1;
            }
        }
        q3k::gemm_q8(L.wo_q, L.wo_s, d->P, d->dim, d->pf_aq, d->pf_as, bc, d->pf_x, d->dim, q3k::EPI_RESID, d->st);
        q3k::rows_quantize(d->pf_x, d->dim, L.ffn_nw, d->dim, bc, d->pf_xq, d->pf_xs, d->st);
        q3k::gemm_q8(L.gu_q, L.gu_s, d->dim, 2 * d->hid, d->pf_xq, d->pf_xs, bc, d->pf_h, d->hid, q3k::EPI_SWIGLU, d->st);
        q3k::rows_quantize(d->pf_h, d->hid, nullptr, d->hid, bc, d->pf_xq, d->pf_xs, d->st);
        q3k::gemm_q8(L.dn_q, L.dn_s, d->hid, d->dim, d->pf_xq, d->pf_xs, bc, d->pf_x, d->dim, q3k::EPI_RESID, d->st);
    }
}
// the same chunk on an fp16-attached model (BASELINE config 5 in its batched, matrix-core form): binary16
// weights x binary16-rounded activation rows on v_mfma_f32_16x16x32_f16, fp32 accumulation; attention is the
// shared fp32 kernel.  No reference arithmetic to match (parity unpinned): checked against orc_forward_f16.
void prefill_chunk_f16(Dev* d, const int* tokens, int bc, int pos0) {
    const int QKV = d->P + 2 * d->KVD;
    HIPCHK(hipMemcpyAsync(d->pf_tokens, tokens, (size_t)bc * sizeof(int), hipMemcpyHostToDevice, d->st));
    q3k::prefill_begin(d->pf_tokens, bc, pos0, nullptr, nullptr, d->dim, d->pf_x, d->dim, d->rope, d->hd, d->pf_cs,
                       d->pf_ctl, d->st);
    q3k::embed_rows_half(d->pf_tokens, bc, d->emb_h, d->dim, d->pf_x, d->dim, d->st);
    for (int l = d->l0; l < d->l1; l++) {
        const LayerDev& L = d->layers[l];
        q3k::rows_half(d->pf_x, d->dim, L.att_nw, d->dim, bc, d->pf_xh, d->st);
        q3k::gemm_f16(L.qkv_h, d->dim, QKV, d->pf_xh, bc, d->pf_qkv, QKV, q3k::EPI_STORE, d->st);
        {
            q3k::Attn a = attn_args(d, l, 0);
            a.ctl = d->pf_ctl; a.qkv = d->pf_qkv; a.cs = d->pf_cs; a.oq = d->pf_aq; a.os = d->pf_as; a.of = d->pf_attf;
            a.part = d->pf_part; a.tickets = d->pf_tickets;
            a.zs_qkv = QKV; a.zs_cs = d->hd; a.zs_oq = d->P; a.zs_os = d->P / 64; a.zs_tickets = d->KV; a.zs_of = d->P;
            a.zs_part = (size_t)d->H * d->max_chunks * (d->hd + 2);
            q3k::kv_append(a, bc, d->st);
            for (int t0 = 0; t0 < bc;) {
                const q3k::AttMode mode = q3k::attn_mode(pos0 + t0);
                int t1 = t0 + 1;
                while (t1 < bc && q3k::attn_mode(pos0 + t1) == mode) t1++;
                q3k::Attn b = a;
                b.ctl += t0; b.qkv += (size_t)t0 * a.zs_qkv; b.cs += (size_t)t0 * a.zs_cs; b.oq += (size_t)t0 * a.zs_oq;
                b.os += (size_t)t0 * a.zs_os; b.part += (size_t)t0 * a.zs_part; b.tickets += (size_t)t0 * a.zs_tickets;
                b.of += (size_t)t0 * a.zs_of;
                b.nz = t1 - t0;
                // the positions are known here (no graph to keep in shape): only the chunk slots the last one needs
                const int need = (pos0 + t1 - 1) / Q3_ATT_CHUNK + 1;
                q3k::attn(b, need < d->chunk_slots ? need : d->chunk_slots, mode, d->st);
                t0 = t1;
            }
        }
        q3k::rows_half(d->pf_attf, d->P, nullptr, d->P, bc, d->pf_xh, d->st);
        q3k::gemm_f16(L.wo_h, d->P, d->dim, d->pf_xh, bc, d->pf_x, d->dim, q3k::EPI_RESID, d->st);
        q3k::rows_half(d->pf_x, d->dim, L.ffn_nw, d->dim, bc, d->pf_xh, d->st);
        q3k::gemm_f16(L.gu_h, d->dim, 2 * d->hid, d->pf_xh, bc, d->pf_h, d->hid, q3k::EPI_SWIGLU, d->st);
        q3k::rows_half(d->pf_h, d->hid, nullptr, d->hid, bc, d->pf_xh, d->st);
        q3k::gemm_f16(L.dn_h, d->hid, d->dim, d->pf_xh, bc, d->pf_x, d->dim, q3k::EPI_RESID, d->st);
    }
}
}  // namespace

/* Prompt ingestion: what completion()'s prompt loop (src/completion.c:57-66) does with n calls of
 * forward(), up to 64 positions at a time on the matrix cores.  Leaves the KV cache and the returned logits
 * (those of the last prompt token, in m->state.logits) bit-identical to the n calls. */
float* q3_prefill(Model* m, const int* tokens, int n, int pos0) {
    Dev* d = attach(m);
    if (d->world > 1) Q3_DIE("q3_prefill: single-GPU models only");
    if (!tokens || n < 1) Q3_DIE("q3_prefill: empty prompt");
    if (pos0 < 0 || pos0 + n > d->seq) Q3_DIE("q3_prefill: positions [%d,%d) outside the context window [0,%d)", pos0, pos0 + n, d->seq);
    for (int i = 0; i < n; i++) {
        if (tokens[i] < 0 || tokens[i] >= d->V) Q3_DIE("q3_prefill: token %d outside the vocabulary [0,%d)", tokens[i], d->V);
    }
    HIPCHK(hipSetDevice(d->device));
    ensure_prefill(d);
    sync_checked(d);                     // steps queued earlier are settled before the passes below wait on the stream
    float* host_logits = m->state.logits;
    const std::vector<int> toks(tokens, tokens + n);
    auto ingest = [d, toks, n, pos0, host_logits] {
        for (int c0 = 0; c0 < n; c0 += Q3_PF_CHUNK) {
            const int bc = n - c0 < Q3_PF_CHUNK ? n - c0 : Q3_PF_CHUNK;
            if (d->fp16) prefill_chunk_f16(d, toks.data() + c0, bc, pos0 + c0);
            else prefill_chunk(d, toks.data() + c0, bc, pos0 + c0);
            if (c0 + bc < n) HIPCHK(hipStreamSynchronize(d->st));      // the token upload buffer is reused
        }
        const int last = (n - 1) % Q3_PF_CHUNK;
        HIPCHK(hipMemcpyAsync(d->x, d->pf_x + (size_t)last * d->dim, (size_t)d->dim * 4, hipMemcpyDeviceToDevice, d->st));
        enqueue_head(d);
        HIPCHK(hipMemcpyAsync(host_logits, d->logits, (size_t)d->V * 4, hipMemcpyDeviceToHost, d->st));
    };
    ingest();
    remember(d, ingest);
    sync_checked(d);
    return m->state.logits;
}

// ---- device-side sampling ---------------------------------------------
namespace {
void ensure_sampler(Dev* d) {
    if (d->sb_ready) return;
    d->sb.pmax = dalloc<float>(d, Q3_SAMPLE_MAX_CHUNKS);
    d->sb.psum = dalloc<float>(d, Q3_SAMPLE_MAX_CHUNKS);
    d->sb.idx_in = dalloc<int>(d, d->V);
    d->sb.key_out = dalloc<float>(d, d->V);
    d->sb.idx_out = dalloc<int>(d, d->V);
    d->sb.tmp_bytes = q3k::sample_temp_bytes(d->V);
    d->sb.tmp = dalloc<char>(d, d->sb.tmp_bytes ? d->sb.tmp_bytes : 16);
    d->seed_dev = dalloc<unsigned long long>(d, 1);
    d->samp_tok = dalloc<int>(d, 1);
    HIPCHK(hipHostMalloc((void**)&d->samp_tok_host, sizeof(int), hipHostMallocDefault));
    q3k::sample_init(d->sb, d->V, d->st);
    d->sb_ready = true;
}
// the clamps of the reference's sampler_create() (src/sampler.c:33-52)
void clamp_sampler(float& temperature, float& top_p) {
    const float epsilon = 1e-6f;
    if (top_p > 1.0f || std::isnan(top_p) || (std::isinf(top_p) && top_p > 0)) top_p = 1.0f;
    else if (top_p < epsilon) top_p = epsilon;
    if (std::isnan(temperature) || (std::isinf(temperature) && temperature > 0)) temperature = 1.0f;
    else if (temperature < epsilon) temperature = epsilon;
}
// reference src/xorshift.c:7-16
float host_xorshift_float(uint64_t* state) {
    *state ^= *state >> 12;
    *state ^= *state << 25;
    *state ^= *state >> 27;
    const uint32_t r = (uint32_t)((*state * 0x2545F4914F6CDD1Dull) >> 32);
    return (float)(r >> 8) / 16777216.0f;
}
}  // namespace

/* reference sample() (src/sampler.c:189-201) on the logits the last step left on the device
 * (q3_forward_device / forward): temperature, softmax, top-p, one xorshift64* draw from *seed.
 * The device logits are overwritten with the probabilities, as sample() does to its argument. */
int q3_device_sample(Model* m, float temperature, float top_p, uint64_t* seed) {
    Dev* d = attach(m);
    if (!d->has_cls) Q3_DIE("q3_device_sample: this pipeline stage holds no logits");
    if (!seed) Q3_DIE("q3_device_sample: seed is NULL");
    ensure_sampler(d);
    clamp_sampler(temperature, top_p);
    const float coin = host_xorshift_float(seed);
    auto draw = [d, temperature, top_p, coin] {
        q3k::sample(d->logits, d->V, temperature, top_p, coin, nullptr, d->sb, d->samp_tok, nullptr, d->st);
        HIPCHK(hipMemcpyAsync(d->samp_tok_host, d->samp_tok, sizeof(int), hipMemcpyDeviceToHost, d->st));
    };
    draw();
    remember(d, draw);
    sync_checked(d);
    return *d->samp_tok_host;
}

/* q3_generate_greedy with the sampler in place of the argmax: no host round trip per token, the
 * RNG state travels to the device and comes back in *seed. */
int q3_generate_sampled(Model* m, int token, int pos, int n, float temperature, float top_p, uint64_t* seed,
                        int* out_tokens) {
    Dev* d = attach(m);
    if (d->world > 1) Q3_DIE("q3_generate_sampled: single-GPU models only");
    if (!seed) Q3_DIE("q3_generate_sampled: seed is NULL");
    if (pos + n > d->seq) n = d->seq - pos;
    if (n <= 0) return 0;
    ensure_sampler(d);
    clamp_sampler(temperature, top_p);
    unsigned long long s = *seed;
    auto put_seed = [d, s] {
        unsigned long long v = s;
        HIPCHK(hipMemcpyAsync(d->seed_dev, &v, sizeof(v), hipMemcpyHostToDevice, d->st));
        HIPCHK(hipStreamSynchronize(d->st));
    };
    sync_checked(d);                     // (what was queued before is settled first: the seed upload below waits on the stream)
    put_seed();
    remember(d, put_seed);
    d->samp_on = true; d->samp_t = temperature; d->samp_p = top_p;
    pipeline_run(d, token, pos, n);
    d->samp_on = false;
    if (out_tokens) q3_pipeline_tokens(m, 0, out_tokens, n);
    sync_checked(d);
    HIPCHK(hipMemcpy(&s, d->seed_dev, sizeof(s), hipMemcpyDeviceToHost));
    *seed = s;
    return n;
}

/* Device-to-device copy rate of this GPU in GB/s, counting the bytes read AND the bytes written
 * (the roofline is quoted against the vendor peak and against a measured copy). */
double q3_measure_copy_gbps(size_t bytes, int iters) {
    die_if_no_gpu();
    HIPCHK(hipSetDevice(pick_device()));
    if (bytes < (1u << 20)) bytes = 1u << 20;
    if (iters < 1) iters = 1;
    void *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc(&a, bytes));
    HIPCHK(hipMalloc(&b, bytes));
    HIPCHK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice));          // warm-up
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++) HIPCHK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    HIPCHK(hipFree(a));
    HIPCHK(hipFree(b));
    return 2.0 * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
}

/* The token loop of the reference's completion() (src/completion.c:57-84) without the tokenizer: the
 * prompt ids go in through q3_prefill, then tokens are sampled on the device until `max_new` of them
 * exist, the context window is full, or the sampler returns one of the two stop ids (the reference
 * stops on BOS / EOS, which it does not emit).  Generation runs in blocks of up to 16 steps without
 * a host round trip; whatever a block computed past the end is discarded, and *seed comes back
 * advanced by exactly the draws the reference loop would have made, so the call is interchangeable
 * with that loop for the same Sampler state.  Returns the number of tokens written to `out`. */
int q3_complete(Model* m, const int* ids, int n_ids, float temperature, float top_p, uint64_t* seed, int stop_a, int stop_b,
                int* out, int max_new) {
    Dev* d = attach(m);
    if (d->world > 1) Q3_DIE("q3_complete: single-GPU models only");
    if (!ids || n_ids < 1 || !seed || (max_new > 0 && !out)) Q3_DIE("q3_complete: bad arguments");
    if (n_ids > d->seq) Q3_DIE("q3_complete: prompt of %d tokens exceeds the context window %d", n_ids, d->seq);
    // the reference checks `next` against the stop ids while it is still feeding the prompt
    for (int k = 1; k < n_ids; k++) {
        if (ids[k] == stop_a || ids[k] == stop_b) {
            q3_prefill(m, ids, k, 0);
            return 0;
        }
    }
    q3_prefill(m, ids, n_ids, 0);
    const uint64_t seed0 = *seed;
    uint64_t sgen = seed0;               // RNG state of the candidate generator (may run ahead)
    std::vector<int> cand;               // cand[i-1] = c_i, the i-th token the sampler returns
    int produced = 0, i = 0;             // i = draws the reference loop has made so far
    while (produced < max_new) {
        if (i == (int)cand.size()) {
            if (i == 0) {                // c_1: from the logits of the last prompt position
                cand.push_back(q3_device_sample(m, temperature, top_p, &sgen));
            } else {                     // feed c_i at position n_ids-1+i, and go on from there on the device
                const int p0 = n_ids - 1 + i;
                int block = d->seq - p0 < 16 ? d->seq - p0 : 16;
                if (block > max_new - produced) block = max_new - produced;
                std::vector<int> got((size_t)block);
                q3_generate_sampled(m, cand[(size_t)i - 1], p0, block, temperature, top_p, &sgen, got.data());
                cand.insert(cand.end(), got.begin(), got.end());
            }
        }
        const int next = cand[(size_t)i++];
        if (next == stop_a || next == stop_b) break;
        out[produced++] = next;
        if (n_ids - 1 + i >= d->seq) break;           // `next` would be fed beyond the window: the reference's loop bound
    }
    uint64_t s = seed0;
    for (int k = 0; k < i; k++) (void)host_xorshift_float(&s);
    *seed = s;
    return produced;
}

void q3_kv_fill_random(Model* m, int T, uint64_t seed) {
    Dev* d = attach(m);
    if (T > d->seq) T = d->seq;
    for (int l = d->l0; l < d->l1; l++) {
        for (int g = 0; g < d->KV; g++) {
            const size_t off = (size_t)g * d->seq_pad * d->hd;
            q3k::fill_random(d->layers[l].kc + off, (size_t)T * d->hd, seed + 2 * (l * 64 + g), d->st);
            q3k::fill_random(d->layers[l].vc + off, (size_t)T * d->hd, seed + 2 * (l * 64 + g) + 1, d->st);
        }
    }
    HIPCHK(hipStreamSynchronize(d->st));
}

void q3_tap_enable(Model* m, int on) {
    Dev* d = attach(m);
    d->tap = on != 0;
}

const float* q3_tap_data(Model* m) {
    Dev* d = attach(m);
    return d->tap_host.data();
}

void q3_layer_step(Model* m, int layer, int pos, const float* x_in, float* x_out) {
    Dev* d = attach(m);
    if (layer < d->l0 || layer >= d->l1) Q3_DIE("layer %d is not on this device", layer);
    check_step_args(d, 0, pos);
    q3k::set_ctl(d->ctl, nullptr, 0, pos, d->st);
    q3k::begin_step(d->ctl, nullptr, nullptr, d->dim, d->x, d->rope, d->hd, d->cs_cur, d->st, d->epoch);
    HIPCHK(hipMemcpyAsync(d->x, x_in, (size_t)d->dim * 4, hipMemcpyHostToDevice, d->st));
    enqueue_layer(d, layer, q3k::attn_mode(pos), 0, q3k::step_rows_cap(pos));
    HIPCHK(hipStreamSynchronize(d->st));
    if (handoff_failed(d)) {             // redo the one layer with separate launches
        HIPCHK(hipMemcpyAsync(d->x, x_in, (size_t)d->dim * 4, hipMemcpyHostToDevice, d->st));
        enqueue_layer(d, layer, q3k::attn_mode(pos), 0, q3k::step_rows_cap(pos));
        HIPCHK(hipStreamSynchronize(d->st));
    }
    HIPCHK(hipMemcpy(x_out, d->x, (size_t)d->dim * 4, hipMemcpyDeviceToHost));
    prof_collect(d);
}

void q3_prof_enable(Model* m, int on) {
    Dev* d = attach(m);
    d->prof = on != 0;
}

// Reserved: nothing is subtracted from the event brackets.  A bracket spans the kernel plus
// the command processor's end-of-kernel release before the closing marker (~2 us more than
// the dispatch duration rocprofv3 reports); an EMPTY bracket measures something else again
// (~4.6 us), so it cannot serve as a correction.
/* mean duration by the in-kernel device clock (first workgroup in .. last workgroup out) of the
 * launches of kernel class `name`; 0 when the class has no such marks */
double q3_prof_device_us(Model* m, const char* name) {
    Dev* d = attach(m);
    for (auto& s : d->prof_slots) {
        if (!strcmp(s.name, name)) return s.dev_launches ? 1000.0 * s.dev_ms / (double)s.dev_launches : 0.0;
    }
    return 0.0;
}

void q3_prof_reset(Model* m) {
    Dev* d = attach(m);
    for (auto& s : d->prof_slots) {
        s.launches = 0;
        s.ms = 0.0;
        s.dev_ms = 0.0;
        s.dev_launches = 0;
    }
}

int q3_prof_get(Model* m, Q3ProfEntry* out, int max_entries) {
    Dev* d = attach(m);
    int n = 0;
    for (auto& s : d->prof_slots) {
        if (n >= max_entries) break;
        out[n].name = s.name;
        out[n].launches = s.launches;
        out[n].ms_total = s.ms;
        out[n].bytes_per_launch = s.bytes;
        n++;
    }
    return n;
}

// ---- reference symbols on host pointers -------------------------------------

void rmsnorm(float* out, float* x, float* w, int size) {
    hipStream_t st = ops_stream();
    DBuf dx(x, (size_t)size * 4), dw(w, (size_t)size * 4), dout((size_t)size * 4);
    q3k::rmsnorm(dout.as<float>(), dx.as<float>(), dw.as<float>(), size, st);
    dout.to_host(out, st);
}

// The reference's softmax() reports non-finite values on stderr as it goes (src/forward.c:38-67: "[Softmax] Invalid
// input" for x[i], i >= 1, in its max loop; "[Softmax] NaN/Inf at" for every exponential that is not finite).  The
// arithmetic runs on the device; the reports are restated here on the host copy the caller handed in, in the reference's
// order at one thread, and cost one vectorisable scan when every input is finite (the only case a healthy model
// produces: an exponential of a finite difference to the maximum cannot be NaN or Inf).
static void softmax_diagnostics(const float* x, int size) {
    int bad = 0;
    for (int i = 0; i < size; i++) bad |= !(fabsf(x[i]) <= 3.402823466e38f);
    if (!bad) return;
    float max_val = x[0];
    for (int i = 1; i < size; i++) {
        if (std::isnan(x[i]) || std::isinf(x[i])) fprintf(stderr, "[Softmax] Invalid input: x[%d] = %f\n", i, (double)x[i]);
        if (x[i] > max_val) max_val = x[i];
    }
    for (int i = 0; i < size; i++) {
        const float e = expf(x[i] - max_val);
        if (std::isnan(e) || std::isinf(e))
            fprintf(stderr, "[Softmax] NaN/Inf at i=%d: x=%f max_val=%f\n", i, (double)x[i], (double)max_val);
    }
}

void softmax(float* x, int size) {
    softmax_diagnostics(x, size);
    hipStream_t st = ops_stream();
    DBuf dx(x, (size_t)size * 4);
    q3k::softmax(dx.as<float>(), size, st);
    dx.to_host(x, st);
}

void matmul(float* out, Q8Tensor* x, Q8Tensor* w, int n, int d, int block_size) {
    if (block_size != Q3_GROUP) Q3_DIE("matmul: group size %d is not 64", block_size);
    q3_op_gemv(w->q, w->s, x->q, x->s, n, d, out);
}

void rotary(float* x, int head_dim, int pos) {
    hipStream_t st = ops_stream();
    std::vector<float> cs;
    rope_cs_host(head_dim, pos, cs);
    DBuf dx(x, (size_t)head_dim * 4), dcs(cs.data(), cs.size() * 4);
    q3k::rope_pairs(dx.as<float>(), 1, head_dim, dcs.as<float>(), st);
    dx.to_host(x, st);
}

// scalar helpers: the same q3_expf the kernels use (q3_numerics.h), evaluated in place
float sigmoid(float x) { return 1.0f / (1.0f + q3_expf(-x)); }
float silu(float x) { return x * sigmoid(x); }

void swiglu(float* x1, float* x3, int size) {
    hipStream_t st = ops_stream();
    DBuf d1(x1, (size_t)size * 4), d3(x3, (size_t)size * 4);
    q3k::swiglu(d1.as<float>(), d3.as<float>(), size, d1.as<float>(), st);
    d1.to_host(x1, st);
}

void q3_op_attention(const float* q, const float* kcache, const float* vcache, int T, int n_heads,
                     int n_kv_heads, int head_dim, float* out);

/* reference attention() (src/forward.c:141-195): reads s->q (already normed and rotated by the caller) and
 * rows 0..pos of the layer's HOST k/v cache, writes the head outputs to s->x_rms_norm; it neither norms,
 * rotates nor appends to a cache.  The device's own cache and step buffers are not touched. */
void attention(Model* m, int layer, int pos) {
    if (!m) Q3_DIE("attention: NULL model");
    const ModelParams* p = &m->params;
    const ForwardState* st = &m->state;
    if (layer < 0 || layer >= p->n_layers || pos < 0 || pos >= p->seq_len) Q3_DIE("attention: layer %d / pos %d out of range", layer, pos);
    if (!st->q || !st->k_cache || !st->v_cache || !st->x_rms_norm) {
        Q3_DIE("attention(): this Model carries no host-side q / KV cache (opened with Q3_OPEN_DEFAULT); "
               "the exported symbol has the reference's host-state semantics");
    }
    const size_t kvd = (size_t)p->n_kv_heads * p->head_dim;
    const size_t loff = (size_t)layer * p->seq_len * kvd;
    q3_op_attention(st->q, st->k_cache + loff, st->v_cache + loff, pos + 1, p->n_heads, p->n_kv_heads, p->head_dim,
                    st->x_rms_norm);
}

void q8_quantize(Q8Tensor* qt, float* x, int n, int block_size) {
    if (block_size != Q3_GROUP) Q3_DIE("q8_quantize: group size %d is not 64", block_size);
    q3_op_quantize(x, n, qt->q, qt->s);
}

void q8_dequantize(Q8Tensor* qt, float* x, int n, int block_size) {
    if (block_size != Q3_GROUP) Q3_DIE("q8_dequantize: group size %d is not 64", block_size);
    hipStream_t st = ops_stream();
    DBuf dq(qt->q, (size_t)n), ds(qt->s, (size_t)n / 64 * 4), dx((size_t)n * 4);
    q3k::dequantize(dq.as<int8_t>(), ds.as<float>(), n, dx.as<float>(), st);
    dx.to_host(x, st);
}

// ---- op-level hooks ------------------------------------------------------------

void q3_op_quantize(const float* x, int n, int8_t* q, float* s) {
    hipStream_t st = ops_stream();
    DBuf dx(x, (size_t)n * 4), dq((size_t)n), ds((size_t)n / 64 * 4);
    q3k::quantize(dx.as<float>(), n, dq.as<int8_t>(), ds.as<float>(), st);
    dq.to_host(q, st);
    ds.to_host(s, st);
}

void q3_op_rmsnorm_quantize(const float* x, const float* w, int n, float* normed, int8_t* q, float* s) {
    hipStream_t st = ops_stream();
    DBuf dx(x, (size_t)n * 4), dw(w, (size_t)n * 4), dn((size_t)n * 4), dq((size_t)n), ds((size_t)n / 64 * 4);
    q3k::rmsnorm_quantize(dx.as<float>(), dw.as<float>(), n, dn.as<float>(), dq.as<int8_t>(), ds.as<float>(), st);
    dn.to_host(normed, st);
    dq.to_host(q, st);
    ds.to_host(s, st);
}

void q3_op_gemv(const int8_t* wq, const float* ws, const int8_t* xq, const float* xs, int n, int d, float* out) {
    if (n % 64 || d % 2) Q3_DIE("gemv: n must be a multiple of 64 and d even (n=%d d=%d)", n, d);
    hipStream_t st = ops_stream();
    DBuf dw(wq, (size_t)n * d), dws(ws, (size_t)n * d / 64 * 4), dx(xq, (size_t)n), dxs(xs, (size_t)n / 64 * 4);
    DBuf dout((size_t)d * 4);
    q3k::Gemv g;
    memset(&g, 0, sizeof(g));
    g.W = dw.as<int8_t>(); g.S = dws.as<float>(); g.n = n; g.d = d;
    g.xq = dx.as<int8_t>(); g.xs = dxs.as<float>(); g.out = dout.as<float>();
    q3k::gemv(g, q3k::PRO_Q8, q3k::EPI_STORE, st);
    dout.to_host(out, st);
}

// the prefill GEMM (int8 MFMA) on `ntok` <= 32 quantised activation rows: out[t][d]
void q3_op_gemm(const int8_t* wq, const float* ws, const int8_t* xq, const float* xs, int n, int d, int ntok, float* out) {
    if (n % 64 || d % 2 || ntok < 1 || ntok > 64) Q3_DIE("gemm: bad shape (n=%d d=%d tokens=%d)", n, d, ntok);
    hipStream_t st = ops_stream();
    DBuf dw(wq, (size_t)n * d), dws(ws, (size_t)n * d / 64 * 4), dx(xq, (size_t)n * ntok), dxs(xs, (size_t)n / 64 * 4 * ntok);
    DBuf dout((size_t)d * 4 * ntok);
    q3k::gemm_q8(dw.as<int8_t>(), dws.as<float>(), n, d, dx.as<int8_t>(), dxs.as<float>(), ntok, dout.as<float>(), d,
                 q3k::EPI_STORE, st);
    dout.to_host(out, st);
}

void q3_op_headnorm_rope(float* heads, int n_heads, int head_dim, const float* w, int pos) {
    hipStream_t st = ops_stream();
    std::vector<float> cs;
    rope_cs_host(head_dim, pos, cs);
    DBuf dh(heads, (size_t)n_heads * head_dim * 4), dw(w, (size_t)head_dim * 4), dcs(cs.data(), cs.size() * 4);
    q3k::headnorm_rope(dh.as<float>(), n_heads, head_dim, dw.as<float>(), dcs.as<float>(), st);
    dh.to_host(heads, st);
}

void q3_op_attention(const float* q, const float* kcache, const float* vcache, int T, int n_heads,
                     int n_kv_heads, int head_dim, float* out) {
    // Runs the production attention kernel in its `prepared` mode: q and the k/v of the
    // last position are taken as given (already normed + rotated), the cache holds 0..T-2.
    hipStream_t st = ops_stream();
    if (T < 1) Q3_DIE("attention: T must be >= 1");
    const int hd = head_dim, P = n_heads * hd, KVD = n_kv_heads * hd;
    const int seq = (T + Q3_ATT_CHUNK - 1) / Q3_ATT_CHUNK * Q3_ATT_CHUNK;
    // cache in device layout [n_kv][seq][hd]
    std::vector<float> kc((size_t)KVD * seq), vc((size_t)KVD * seq);
    for (int t = 0; t < T; t++) {
        for (int g = 0; g < n_kv_heads; g++) {
            memcpy(&kc[((size_t)g * seq + t) * hd], kcache + ((size_t)t * n_kv_heads + g) * hd, (size_t)hd * 4);
            memcpy(&vc[((size_t)g * seq + t) * hd], vcache + ((size_t)t * n_kv_heads + g) * hd, (size_t)hd * 4);
        }
    }
    std::vector<float> qkv((size_t)P + 2 * KVD);
    memcpy(qkv.data(), q, (size_t)P * 4);
    memcpy(qkv.data() + P, kcache + (size_t)(T - 1) * KVD, (size_t)KVD * 4);
    memcpy(qkv.data() + P + KVD, vcache + (size_t)(T - 1) * KVD, (size_t)KVD * 4);
    const int max_chunks = (T + Q3_ATT_CHUNK - 1) / Q3_ATT_CHUNK;
    DBuf dqkv(qkv.data(), qkv.size() * 4), dkc(kc.data(), kc.size() * 4), dvc(vc.data(), vc.size() * 4);
    DBuf dpart((size_t)n_heads * max_chunks * (hd + 2) * 4), doq((size_t)P), dos((size_t)P / 64 * 4 + 16);
    DBuf dtick((size_t)n_kv_heads * 4);
    HIPCHK(hipMemset(dtick.p, 0, dtick.bytes));
    DBuf dof((size_t)P * 4);
    q3k::Ctl ctl = {0, T - 1};
    DBuf dctl(&ctl, sizeof(ctl));
    q3k::Attn a;
    memset(&a, 0, sizeof(a));
    a.ctl = dctl.as<q3k::Ctl>(); a.qkv = dqkv.as<float>(); a.qnw = nullptr; a.knw = nullptr; a.cs = nullptr;
    a.kc = dkc.as<float>(); a.vc = dvc.as<float>(); a.part = dpart.as<float>(); a.tickets = dtick.as<unsigned>();
    a.oq = doq.as<int8_t>(); a.os = dos.as<float>(); a.of = dof.as<float>(); a.qdbg = nullptr;
    a.n_heads = n_heads; a.n_kv = n_kv_heads; a.hd = hd; a.seq_len = seq; a.max_chunks = max_chunks;
    a.prepared = 1;
    q3k::attn(a, max_chunks < 64 ? max_chunks : 64, q3k::attn_mode(T - 1), st);
    dof.to_host(out, st);
}

void q3_op_swiglu(const float* gate, const float* up, int n, float* out) {
    hipStream_t st = ops_stream();
    DBuf dg(gate, (size_t)n * 4), du(up, (size_t)n * 4), dout((size_t)n * 4);
    q3k::swiglu(dg.as<float>(), du.as<float>(), n, dout.as<float>(), st);
    dout.to_host(out, st);
}

// sample() on host logits: returns the token, leaves the probabilities in `logits` (as the
// reference does) and advances *seed by one draw
int q3_op_sample(float* logits, int n, float temperature, float top_p, uint64_t* seed) {
    hipStream_t st = ops_stream();
    if (n < 1 || !seed) Q3_DIE("q3_op_sample: bad arguments");
    clamp_sampler(temperature, top_p);
    DBuf dl(logits, (size_t)n * 4), dmax((size_t)Q3_SAMPLE_MAX_CHUNKS * 4), dsum((size_t)Q3_SAMPLE_MAX_CHUNKS * 4);
    DBuf didx((size_t)n * 4), dkey((size_t)n * 4), dido((size_t)n * 4), dtok(sizeof(int));
    q3k::SampleBufs b;
    b.pmax = dmax.as<float>(); b.psum = dsum.as<float>(); b.idx_in = didx.as<int>();
    b.key_out = dkey.as<float>(); b.idx_out = dido.as<int>();
    b.tmp_bytes = q3k::sample_temp_bytes(n);
    DBuf dtmp(b.tmp_bytes ? b.tmp_bytes : 16);
    b.tmp = dtmp.p;
    q3k::sample_init(b, n, st);
    const float coin = host_xorshift_float(seed);
    q3k::sample(dl.as<float>(), n, temperature, top_p, coin, nullptr, b, dtok.as<int>(), nullptr, st);
    int tok = -1;
    dl.to_host(logits, st);
    dtok.to_host(&tok, st);
    return tok;
}

void q3_op_expf(const float* x, int n, float* out) {
    hipStream_t st = ops_stream();
    DBuf dx(x, (size_t)n * 4), dout((size_t)n * 4);
    q3k::expf_map(dx.as<float>(), n, dout.as<float>(), st);
    dout.to_host(out, st);
}

// ---- pipeline (filled in by q3_pipeline.hip-style code below) --------------------

// Self-test of the pipeline code on ONE GPU: `world` stages of the same checkpoint live in
// this process (each uploads only its own layers), ticks are executed in the order the
// schedule prescribes, and the RCCL hand-offs are replaced by stream-ordered device
// copies.  Everything else -- layer split, per-stream KV caches, tick schedule, token
// feedback, graphs -- is the code the multi-process run executes.
// out_tokens[world][nsteps]; returns 0 on success.
int q3_pipeline_selftest_streams(const char* path, int seq_len, int world, int streams, int first_token, int pos0,
                                 int nsteps, int* out_tokens);
int q3_pipeline_selftest(const char* path, int seq_len, int world, int first_token, int pos0, int nsteps,
                         int* out_tokens) {
    return q3_pipeline_selftest_streams(path, seq_len, world, 0, first_token, pos0, nsteps, out_tokens);
}
// the same with only the first `streams` streams running (0 = all): out_tokens[streams][nsteps]
int q3_pipeline_selftest_streams(const char* path, int seq_len, int world, int streams, int first_token, int pos0,
                                 int nsteps, int* out_tokens) {
    if (world < 1 || world > 64) return -1;
    if (streams <= 0 || streams > world) streams = world;
    std::vector<Model*> ms(world, nullptr);
    std::vector<Dev*> ds(world, nullptr);
    for (int r = 0; r < world; r++) {
        ms[r] = q3_model_open(path, seq_len, 0);
        if (!ms[r]) return -1;
        AttachOpts o;
        o.stage_rank = r;
        o.stage_world = world;
        ds[r] = attach(ms[r], o);
    }
    for (int r = 0; r < world; r++) {
        ds[r]->loop_prev = ds[(r + world - 1) % world];
        // one stream for all stages: program order = tick order = data dependencies
        if (r > 0) {
            HIPCHK(hipStreamDestroy(ds[r]->st));
            ds[r]->st = ds[0]->st;
        }
        pipeline_check(ds[r], first_token, pos0, nsteps);
    }
    const int T = nsteps * world + world - 1;
    for (int t = 0; t < T; t++) {
        for (int r = 0; r < world; r++) {
            int s = 0, k = 0;
            if (q3_pipeline_schedule(r, world, nsteps, t, &s, &k) && s < streams) pipeline_tick(ds[r], first_token, pos0, s, k);
        }
        if (t < T - 1) {
            for (int r = 0; r < world; r++) ring_exchange(ds[r], t);     // sends
            for (int r = 0; r < world; r++) loopback_deliver(ds[r], t);  // receives
        }
    }
    HIPCHK(hipStreamSynchronize(ds[0]->st));
    for (int s = 0; s < streams; s++) q3_pipeline_tokens(ms[world - 1], s, out_tokens + (size_t)s * nsteps, nsteps);
    for (int r = world - 1; r >= 0; r--) {
        if (r > 0) HIPCHK(hipStreamCreateWithFlags(&ds[r]->st, hipStreamNonBlocking));   // detach destroys it
        q3_model_close(ms[r]);
    }
    return 0;
}

int q3_pipeline_unique_id(void* id_bytes) {
    setenv("NCCL_SOCKET_IFNAME", "lo", 0);   // single-node pipeline: bootstrap over loopback
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
    memset(id_bytes, 0, Q3_PIPE_ID_BYTES);
    memcpy(id_bytes, &id, sizeof(id) < Q3_PIPE_ID_BYTES ? sizeof(id) : Q3_PIPE_ID_BYTES);
    return 0;
}

/* ranks of the RCCL communicator the pipeline runs on (1 = no pipeline) */
int q3_pipeline_size(void) {
    if (!(g_pipe.on || g_pipe.self) || !g_pipe.comm) return 1;
    int n = 0;
    if (ncclCommCount(g_pipe.comm, &n) != ncclSuccess) return -1;
    return n;
}

int q3_pipeline_init(int rank, int world, const void* id_bytes) {
    die_if_no_gpu();
    if (world < 1 || rank < 0 || rank >= world) return -1;
    {
        // one process per GPU: two ranks on one device cannot form an RCCL communicator
        int ndev = 0;
        HIPCHK(hipGetDeviceCount(&ndev));
        if (world > ndev && !getenv("Q3_DEVICE")) {
            Q3_DIE("pipeline of %d ranks needs %d GPUs, %d visible (one process per GPU)", world, world, ndev);
        }
    }
    HIPCHK(hipSetDevice(pick_device()));
    g_pipe.rank = rank;
    g_pipe.world = world;
    g_pipe.on = world > 1;
    g_pipe.self = world == 1 && getenv("Q3_PIPE_SELF") && atoi(getenv("Q3_PIPE_SELF")) != 0;
    if (world > 1 || g_pipe.self) {
        setenv("NCCL_SOCKET_IFNAME", "lo", 0);
        ncclUniqueId id;
        memcpy(&id, id_bytes, sizeof(id));
        ncclResult_t r = ncclCommInitRank(&g_pipe.comm, world, id, rank);
        if (r != ncclSuccess) {
            fprintf(stderr, "[q3hip] ncclCommInitRank failed: %s\n", ncclGetErrorString(r));
            return -1;
        }
    }
    return 0;
}

// max over ranks of a host double (also serves as a barrier); identity on one rank
double q3_pipeline_allreduce_max(double v) {
    if (!g_pipe.on || g_pipe.world <= 1) return v;
    double* dv = nullptr;
    HIPCHK(hipMalloc((void**)&dv, sizeof(double)));
    HIPCHK(hipMemcpy(dv, &v, sizeof(double), hipMemcpyHostToDevice));
    ncclResult_t r = ncclAllReduce(dv, dv, 1, ncclFloat64, ncclMax, g_pipe.comm, nullptr);
    if (r != ncclSuccess) Q3_DIE("ncclAllReduce failed: %s", ncclGetErrorString(r));
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(hipMemcpy(&v, dv, sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipFree(dv));
    return v;
}

void q3_pipeline_shutdown(void) {
    if (g_pipe.comm) {
        ncclCommDestroy(g_pipe.comm);
        g_pipe.comm = nullptr;
    }
    g_pipe.on = false;
    g_pipe.self = false;
    g_pipe.world = 1;
    g_pipe.rank = 0;
}

}  // extern "C"
