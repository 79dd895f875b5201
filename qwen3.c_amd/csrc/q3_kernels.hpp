// q3_kernels.hpp -- launch interface between the C-ABI shim and the gfx950 kernels.
// Every function enqueues on `st` and returns; no allocation, no synchronisation
// (so the whole step can be captured into a hipGraph).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q3k {

// control block kept in device memory so that a captured graph never needs
// its kernel arguments patched: {token, pos}
struct Ctl {
    int token;
    int pos;
};

enum Pro { PRO_Q8 = 0, PRO_NORM = 1, PRO_F32 = 2 };
enum Epi { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

struct Gemv {
    const int8_t* W;   // [d][n] int8 codes, row-major (reference layout, model.c:131)
    const float* S;    // [d][n/64] group scales
    int n, d;
    // prologue inputs (which are read depends on Pro)
    const int8_t* xq;  // PRO_Q8: activation codes [n]
    const float* xs;   // PRO_Q8: activation scales [n/64]
    const float* xf;   // PRO_NORM / PRO_F32: fp32 activation [n]
    const float* nw;   // PRO_NORM: RMSNorm weight [n]
    // epilogue output: EPI_STORE out[d]; EPI_RESID out[d] += ; EPI_SWIGLU out[d/2]
    float* out;
    // profiling only (else null): clk[0] <- min over workgroups of s_memrealtime at entry,
    // clk[1] <- max at exit (100 MHz device clock)
    unsigned long long* clk;
    // diagnostic builds only (-DQ3_GEMV_STAMPS): phase marks of wave 0 of three workgroups
    unsigned long long* stamps;
};

// Compute units of the device the launches go to (set once per process from hipDeviceProp at the first attach; 256 on
// MI355X).  Launch plans and the fused launches' grids are sized from it.
void set_cu_count(int n);
int cu_count();

void gemv(const Gemv& g, Pro pro, Epi epi, hipStream_t st);

struct Attn {
    const Ctl* ctl;      // pos is read on the device
    const float* qkv;    // raw projections of this step: q[P] | k[KVD] | v[KVD]
    const float* qnw;    // per-layer q head-norm weight [hd]
    const float* knw;    // per-layer k head-norm weight [hd]
    const float* cs;     // (cos, sin) row of this step's position, [hd/2][2] (written by begin_step)
    float* kc;           // this layer's K cache [n_kv][seq_len][hd], seq_len a multiple of 64
    float* vc;           // this layer's V cache, same layout
    float* part;         // chunk partials [n_heads][max_chunks][hd+2]
    unsigned* tickets;   // [n_kv] arrival tickets of the chunk workgroups (zero between launches)
    int8_t* oq;          // attention output codes [P]
    float* os;           // attention output scales [P/64]
    float* of;           // optional fp32 copy of the head outputs [P] (may be null)
    float* qdbg;         // optional: normed+rotated q [P] (may be null)
    int n_heads, n_kv, hd, seq_len, max_chunks;
    unsigned long long* stamps;  // diagnostic builds only: s_memrealtime/s_memtime marks of workgroup (0,0)
    int prepared;        // op-level test hook only: q and the k/v of `pos` are already normed + rotated
    // Batched prompt ingestion: one launch serves nz consecutive positions (grid.z); position z uses
    // ctl[z], qkv + z*zs_qkv, cs + z*zs_cs, oq + z*zs_oq, os + z*zs_os, part + z*zs_part, tickets + z*zs_tickets.
    // Their k/v rows must already be in the cache (kv_append): a later position reads an earlier one's.
    int nz;              // 0 or 1 = a single position
    int zs_qkv, zs_cs, zs_oq, zs_os, zs_tickets, zs_of;
    size_t zs_part;
    // k_attn_wo only: the outputs leave as 8-byte {tag, value} granules for the consumer workgroups of the same
    // launch -- granule i < P/4 carries dword i of the codes, granule P/4 + g the scale of group g; tag =
    // (*epoch << 8) | layer_tag, *epoch being the step counter k_begin advances (so a granule of an earlier step or
    // of another layer never matches)
    unsigned long long* og;
    const unsigned* epoch;
    unsigned layer_tag;
    // k_attn_wo, in-launch-merge shape: the chunk partials as granules too, [n_heads][max_chunks][hd+2] -- the workgroup
    // of chunk slot 0 merges a kv head's chunks as soon as their granules carry the tag (no drain, no ticket)
    unsigned long long* pg;
    int v_hold;          // k_attn_long: device-clock ticks (10 ns) the loader waves hold their V-tile requests back behind the K tile's
};
// k (head norm + RoPE) and v of `ntok` consecutive positions into the cache (reference forward.c:270-286),
// ahead of a batched attn(): qkv rows of stride zs_qkv, (cos,sin) rows of stride hd, positions from ctl[t].pos
void kv_append(const Attn& a, int ntok, hipStream_t st);
// `chunk_slots` workgroups per kv head walk the 64-position chunks of [0,pos].  Three launch
// shapes, chosen by the position (the graph of a step is built once per shape):
//   ATT_SINGLE  pos < 64: one chunk, finalised directly (chunk_slots = 1)
//   ATT_MERGE   each workgroup publishes its chunk partials and the one that draws the last
//               ticket of its kv head merges them -- inside the same launch
//   ATT_LONG    pos >= Q3_ATT_LONG: the partials are merged by a second, wide launch (one wave
//               per 64 output values); a single last arriver would have to pull every
//               partial of its kv head (135 KB at 4096 positions) through one CU
enum AttMode { ATT_SINGLE = 0, ATT_MERGE = 1, ATT_LONG = 2 };
#define Q3_ATT_LONG 1024
inline AttMode attn_mode(int pos) { return pos < 64 ? ATT_SINGLE : (pos < Q3_ATT_LONG ? ATT_MERGE : ATT_LONG); }
// A step's launches are captured once per SHAPE: ATT_SINGLE comes in four, by the rows of the one K/V tile
// that can hold cached positions (pos < 16, 32, 48, 64 -> rows_cap 16, 32, 48, 64: the rest is not requested);
// ATT_LONG in one per 1024 positions, by the chunk slots that hold cached positions for certain (16 per 1024
// positions reached: their workgroups request K/V without waiting for the position to arrive).
inline int step_shape(int pos) { return pos < 64 ? pos / 16 : (pos < Q3_ATT_LONG ? 4 : 4 + pos / Q3_ATT_LONG); }
inline int step_shapes(int seq_len) { return seq_len <= Q3_ATT_LONG ? 5 : 5 + (seq_len - 1) / Q3_ATT_LONG; }
// what attn() gets as `rows_cap`: rows of the tile (one-chunk shapes), 64 (ATT_MERGE), sure chunk slots (ATT_LONG)
inline int step_rows_cap(int pos) { return pos < 64 ? (pos / 16 + 1) * 16 : (pos < Q3_ATT_LONG ? 64 : (pos / Q3_ATT_LONG) * (Q3_ATT_LONG / 64)); }
// a position of shape k (the first one)
inline int step_shape_pos(int k) { return k < 4 ? 16 * k : (k == 4 ? 64 : (k - 4) * Q3_ATT_LONG); }

// The Wo GEMV + residual add behind the attention, in the SAME launch (k_attn_wo in q3_attn.hip; reference
// forward.c:291-298).
struct WoView {
    const int8_t* W;      // [d][n] codes
    const float* S;       // [d][n/64] scales
    int n, d, rpw;        // rows per consumer workgroup
    float* x;             // x[d] += W . att
    const unsigned long long* gran;   // the attention output as {tag, value} granules (Attn::og of the same launch)
    const unsigned* epoch;            // step counter (k_begin)
    unsigned layer_tag;               // low byte of the tag
    unsigned* err;        // host-mapped word, set to 1 when a bounded wait gives up
    unsigned long long* stamps;   // diagnostic builds only (-DQ3_ATTN_STAMPS): eight device-clock marks per consumer workgroup
    int delay;            // device-clock ticks (10 ns) the extra workgroups hold their weight requests back after entry
    unsigned long long wait_ticks;   // a consumer gives up after this much device time (10-ns ticks; 5 s unless Q3_WAIT_TICKS says otherwise: tests)
};
// rows per consumer workgroup / whether the fused launch covers this shape (else: attn() then gemv())
bool attn_wo_supported(const Attn& a, const WoView& w, int chunk_slots, AttMode mode);
// `rows_cap`: see step_rows_cap (one-chunk shapes: tile rows; ATT_LONG: chunk slots certain to exist); 0 = not said: whole
// tiles, and in ATT_LONG the 16 slots that the mode itself implies (at least Q3_ATT_LONG positions are cached).  `wo` non-null: Wo rides along -- with the attention launch
// (ATT_SINGLE / ATT_MERGE: k_attn_wo) or with the merge launch (ATT_LONG: k_merge_wo).
void attn(const Attn& a, int chunk_slots, AttMode mode, hipStream_t st, int rows_cap = 0, const WoView* wo = nullptr);

void embed(const Ctl* ctl, const int8_t* eq, const float* es, int dim, float* x, hipStream_t st);
// first kernel of a step: x = embedding row of ctl->token (eq may be null on later pipeline
// stages) and cs = rope[ctl->pos]
void begin_step(const Ctl* ctl, const int8_t* eq, const float* es, int dim, float* x, const float* rope,
                int hd, float* cs, hipStream_t st, unsigned* epoch = nullptr, int vocab = 0x7fffffff);
// (`vocab`: rows of the embedding table; a token id outside [0, vocab) fetches row 0)

// scratch: 256 words of device memory
void argmax(const float* logits, int n, float* scratch, int* out, int* out2, hipStream_t st);
void set_ctl(Ctl* ctl, const int* tok_src, int tok_imm, int pos, hipStream_t st);

// stand-alone ops behind the reference's exported symbols / the op-level tests
void rmsnorm(float* out, const float* x, const float* w, int n, hipStream_t st);
void softmax(float* x, int n, hipStream_t st);
void quantize(const float* x, int n, int8_t* q, float* s, hipStream_t st);
void rmsnorm_quantize(const float* x, const float* w, int n, float* normed, int8_t* q, float* s,
                      hipStream_t st);
void dequantize(const int8_t* q, const float* s, int n, float* x, hipStream_t st);
void rope_pairs(float* x, int n_heads, int hd, const float* cs /* [hd/2][2] */, hipStream_t st);
void headnorm_rope(float* heads, int n_heads, int hd, const float* w, const float* cs, hipStream_t st);
void swiglu(const float* g, const float* u, int n, float* out, hipStream_t st);
void expf_map(const float* x, int n, float* out, hipStream_t st);
void fill_random(float* p, size_t n, uint64_t seed, hipStream_t st);

// ---- batched prompt ingestion (q3_prefill.hip) --------------------------------------------
// (rmsnorm with weight `w` when non-null, then) q8_quantize of `rows` activation rows of n floats
void rows_quantize(const float* x, int ldx, const float* w, int n, int rows, int8_t* q, float* s, hipStream_t st);
// out[t][r] (leading dimension ldo) = W[r][:] . x_t for t < ntok <= 64, on int8 MFMA, bit-identical to
// gemv() per token; epilogues as in gemv()
void gemm_q8(const int8_t* W, const float* S, int n, int d, const int8_t* xq, const float* xs, int ntok, float* out,
             int ldo, Epi epi, hipStream_t st);
void prefill_begin(const int* tokens, int ntok, int pos0, const int8_t* eq, const float* es, int dim, float* x, int ldx,
                   const float* rope, int hd, float* cs, Ctl* ctl, hipStream_t st);

// ---- fp16 contrast path (q3_fp16.hip; BASELINE config 5) ------------------------------------
void to_half(const int8_t* q, const float* s, size_t n, void* out, hipStream_t st);     // half(q*s)
void embed_half(const Ctl* ctl, const void* e, int dim, float* x, hipStream_t st, int vocab = 0x7fffffff);
// out (=, or += when nw is null) W x, W [d][n] binary16, x fp32 (rmsnorm'ed with weight nw when given);
// EPI_SWIGLU: rows interleaved (gate, up) -> out[d/2].  Only the three combinations a layer uses exist:
// (nw, STORE), (nw, SWIGLU), (null, RESID).
void gemv_f16(const void* W, int n, int d, const float* x, const float* nw, float* out, Epi epi, hipStream_t st);
// batched form on v_mfma_f32_16x16x32_f16 (prompt ingestion of an fp16-attached model): activation rows to
// binary16 ((rmsnorm with w when given) then round), out[t][r] (+)= W[r][:] . X[t][:] for ntok <= 64
void rows_half(const float* x, int ldx, const float* w, int n, int rows, void* out, hipStream_t st);
void gemm_f16(const void* W, int n, int d, const void* X, int ntok, float* out, int ldo, Epi epi, hipStream_t st);
void embed_rows_half(const int* tokens, int ntok, const void* e, int dim, float* x, int ldx, hipStream_t st);

// ---- device-side sampling (q3_sample.hip) -----------------------------------------------
#define Q3_SAMPLE_MAX_CHUNKS 1024
struct SampleBufs {
    float* pmax;      // [Q3_SAMPLE_MAX_CHUNKS]
    float* psum;      // [Q3_SAMPLE_MAX_CHUNKS]
    int* idx_in;      // [n] 0..n-1 (sample_init)
    float* key_out;   // [n] sorted probabilities
    int* idx_out;     // [n] their token ids
    void* tmp;        // radix-sort scratch, sample_temp_bytes(n)
    size_t tmp_bytes;
};
size_t sample_temp_bytes(int n);
void sample_init(const SampleBufs& b, int n, hipStream_t st);
// reference sample() (src/sampler.c:189-201) on the device.  `logits` is overwritten with the
// probabilities, as the reference does.  The coin is `coin` (drawn by the caller) unless
// `seed_dev` is given: then the kernel draws it from that xorshift64* state and advances it.
// The token goes to *out (and *out2 when non-null).
void sample(float* logits, int n, float temperature, float top_p, float coin, unsigned long long* seed_dev,
            const SampleBufs& b, int* out, int* out2, hipStream_t st);


}  // namespace q3k
