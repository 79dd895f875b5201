/*
 * q3_ext.h -- entry points of libq3hip.so beyond the reference's own symbols.
 *
 * The reference has no loader-independent way to build a Model, no synthetic
 * checkpoints, no timers and no multi-GPU knob; everything a harness needs
 * around forward() lives here, behind plain C types.
 *
 * Host-side pieces (plain C, no GPU needed):
 *   q3_model_open/close   own reader of the reference's `.bin` layout
 *                         (reference src/model.c:451-500, format Appendix A of SURVEY.md)
 *   q3_synth_write        deterministic random-init checkpoint writer
 *                         (the exporter qwen3/weights.py:249-381 defines the format)
 *   q3_argmax             greedy pick over host logits
 *
 * Device-side pieces (need an MI355X; fail loudly otherwise):
 *   q3_device_*           explicit attach/detach of the per-Model device state
 *                         that forward() otherwise creates lazily
 *   q3_op_*               single kernels on host arrays, for op-level parity tests
 *   q3_tap_*              per-layer residual read-back for layer-level parity
 *   q3_prof_*             HIP-event timing of the GEMV kernels on the launch stream
 *   q3_generate_greedy    on-device greedy loop (no host round trip per token)
 *   q3_pipeline_*         layer pipeline over several GPUs, one process per GPU
 */
#ifndef Q3_EXT_H
#define Q3_EXT_H

#include "q3_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- host --- */

/* flags for q3_model_open */
#define Q3_OPEN_DEFAULT 0
#define Q3_OPEN_HOST_STATE 1  /* also allocate the reference's host scratch, host KV
                                 cache and fp32 embedding table (model.c:199-206,
                                 321-406) so that CPU code written against the
                                 reference structs (the oracle) can run on it */

/* Map `path`, check the header, carve the weight views.  seq_len is lowered
 * to override_seq_len when 0 < override <= header value (model.c:74-76).
 * state.logits is always allocated (page-aligned).  NULL on failure. */
Model* q3_model_open(const char* path, int override_seq_len, int flags);
void q3_model_close(Model* m);

typedef struct Q3SynthSpec {
    int dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len, head_dim;
    int shared_classifier;
    uint64_t seed;
    float sigma;          /* target std of dequantised weights, 0 -> 0.02 */
} Q3SynthSpec;

/* Fill a spec from a name: "tiny", "small", "0.6B", "1.7B", "4B", "8B".
 * Returns 0, or -1 for an unknown name. */
int q3_synth_preset(const char* name, Q3SynthSpec* spec);
/* Write a random-init checkpoint in the reference's .bin layout.
 * Same spec + seed => same bytes on every machine.  Returns 0 on success. */
int q3_synth_write(const char* path, const Q3SynthSpec* spec);
/* Bytes q3_synth_write will produce for `spec`. */
int64_t q3_synth_bytes(const Q3SynthSpec* spec);
/* Write "<model_path>.tokenizer" in the reference's tokenizer format (src/tokenizer.c:17-120): single bytes, a few
 * merges, Qwen's special strings at the end of the vocabulary, printable fillers elsewhere -- what the reference's
 * qwen_create() / completion() need next to a synthetic checkpoint.  Returns 0 on success. */
int q3_synth_write_tokenizer(const char* model_path, int vocab_size);
/* 64-bit FNV-1a of a file, for fixture checksums. */
uint64_t q3_file_checksum(const char* path);

int q3_argmax(const float* logits, int n);

/* Algorithmic HBM bytes of one decode step with T cached positions before
 * it; and of one GEMV launch of shape d x n. */
double q3_bytes_per_token(const ModelParams* p, int T);
double q3_gemv_bytes(int d, int n);

/* -------------------------------------------------------------- device --- */

/* Number of visible HIP devices (0 when there is no GPU / no driver). */
int q3_device_count(void);

/* Create (or return) the device state of `m`: weights uploaded, KV cache,
 * RoPE table, launch graph.  forward() calls this lazily.  Environment:
 *   Q3_DEVICE=<n>      HIP device ordinal (default: LOCAL_RANK or 0)
 *   Q3_GRAPH=0|1       replay the step through a hipGraph (default 1)
 * Returns 0 on success; on failure prints the reason and returns -1. */
int q3_device_attach(Model* m);
/* The fp16 CONTRAST path (BASELINE config 5; nothing like it in the reference): attach with every Q8_0
 * matrix dequantised (q*s) and rounded to binary16 on the device; forward() then runs on those weights with
 * fp32 activations and no activation quantisation.  Must come before anything else touches the device for
 * this Model.  q3_prefill works on such a Model too (its GEMMs then run on v_mfma_f32_16x16x32_f16 with the
 * activation rows rounded to binary16); the pipeline is Q8_0-only. */
int q3_device_attach_fp16(Model* m);
void q3_device_detach(Model* m);
void q3_device_sync(Model* m);
/* Below 1024 cached positions attention and the Wo GEMV share one launch whose Wo workgroups wait, bounded, for the
 * attention output inside the launch.  Should such a wait ever give up (5 s of device time), the Model switches to
 * separate launches, redoes whatever was queued since its last synchronisation, and goes on; this counts how often. */
int q3_handoff_fallbacks(Model* m);
/* One decode step without the logits copy: logits stay on the device.
 * Pair with q3_logits_fetch() or q3_device_argmax(). */
void q3_forward_device(Model* m, int token, int pos);
void q3_logits_fetch(Model* m);           /* D2H into m->state.logits, synchronous */
int q3_device_argmax(Model* m);           /* argmax of the device logits */

/* Greedy decode on the device: feeds `token` at `pos`, then its own argmax,
 * for n steps; out_tokens[i] is the token chosen after step i.  Logits are
 * copied to the host only after the last step. */
int q3_generate_greedy(Model* m, int token, int pos, int n, int* out_tokens);

/* Device-to-device copy rate of the selected GPU in GB/s (bytes read + bytes written), `iters` copies of
 * `bytes`: the measured companion of the vendor HBM peak in bench.py. */
double q3_measure_copy_gbps(size_t bytes, int iters);

/* The token loop of the reference's completion() (src/completion.c:57-84) on token ids: prompt through
 * q3_prefill, then device-side sampling until `max_new` tokens exist, the context window is full, or the
 * sampler returns stop_a / stop_b (the reference: BOS / EOS; pass -1 for none; not emitted).  Same tokens
 * and same final *seed as the reference loop with a Sampler{temperature, top_p, seed}.  Returns the number
 * of tokens written to `out`. */
int q3_complete(Model* m, const int* ids, int n_ids, float temperature, float top_p, uint64_t* seed, int stop_a, int stop_b,
                int* out, int max_new);

/* Batched prompt ingestion (no counterpart call in the reference, whose completion() feeds the
 * prompt through forward() one token at a time, src/completion.c:57-66): positions pos0..pos0+n-1
 * take `tokens`, 64 at a time, the Q8_0 products on the int8 matrix cores.  The KV cache and the
 * returned logits (of the last prompt token; m->state.logits, host memory) are bit-identical to n
 * calls of forward(). */
float* q3_prefill(Model* m, const int* tokens, int n, int pos0);

/* Device-side sampling (no counterpart call in the reference; same result as its host
 * sample(), src/sampler.c:189-201, applied to these logits: temperature, softmax, top-p
 * nucleus, one xorshift64* draw, src/xorshift.c:7-16).  `temperature` and `top_p` are
 * clamped as sampler_create() does (src/sampler.c:33-52).  The logits the last step left on
 * the device are overwritten with the probabilities, as sample() does to its argument;
 * *seed advances by one draw.  Returns the token. */
int q3_device_sample(Model* m, float temperature, float top_p, uint64_t* seed);
/* q3_generate_greedy() with that sampler in place of the argmax; the RNG state comes back in *seed. */
int q3_generate_sampled(Model* m, int token, int pos, int n, float temperature, float top_p, uint64_t* seed,
                        int* out_tokens);

/* Fill positions [0, T) of every layer's device KV cache with finite
 * pseudo-random values (timing of long contexts without T real steps). */
void q3_kv_fill_random(Model* m, int T, uint64_t seed);

/* Per-layer tap: after q3_tap_enable(m, 1) every forward() also copies the
 * residual after each layer into a host buffer of n_layers*dim floats. */
void q3_tap_enable(Model* m, int on);
const float* q3_tap_data(Model* m);

/* Layer-level teacher forcing: run only layer `layer` at `pos` on the device,
 * starting from the host residual x_in[dim], and return the residual after
 * the layer in x_out[dim].  Uses and updates the device KV cache at `pos`. */
void q3_layer_step(Model* m, int layer, int pos, const float* x_in, float* x_out);

/* ---- single kernels on host arrays (op-level parity; each call uploads,
 *      launches on the library stream and downloads) ------------------- */
void q3_op_quantize(const float* x, int n, int8_t* q, float* s);
void q3_op_rmsnorm_quantize(const float* x, const float* w, int n, float* normed,
                            int8_t* q, float* s);
void q3_op_gemv(const int8_t* wq, const float* ws, const int8_t* xq, const float* xs,
                int n, int d, float* out);
void q3_op_headnorm_rope(float* heads, int n_heads, int head_dim, const float* w, int pos);
/* q[n_heads][hd]; kcache/vcache [T][n_kv][hd] (reference cache layout of one
 * layer, model.c:360-361 / forward.c:148); out[n_heads][hd]. */
void q3_op_attention(const float* q, const float* kcache, const float* vcache, int T,
                     int n_heads, int n_kv_heads, int head_dim, float* out);
void q3_op_swiglu(const float* gate, const float* up, int n, float* out);
void q3_op_expf(const float* x, int n, float* out);
/* The prefill GEMM (int8 MFMA) on ntok <= 64 quantised activation rows xq[ntok][n] / xs[ntok][n/64]:
 * out[t][d] = matmul() of row t (reference src/forward.c:79-101), bit-identical to q3_op_gemv per row. */
void q3_op_gemm(const int8_t* wq, const float* ws, const int8_t* xq, const float* xs, int n, int d, int ntok, float* out);
/* reference sample() (src/sampler.c:189-201) on host logits through the device sampler: returns the
 * token, leaves the probabilities in `logits`, advances *seed by one draw. */
int q3_op_sample(float* logits, int n, float temperature, float top_p, uint64_t* seed);

/* ---- timing ----------------------------------------------------------- */
typedef struct Q3ProfEntry {
    const char* name;     /* kernel class: "qkv", "attn", "wo", "gateup", "down", "cls", ... */
    int64_t launches;
    double ms_total;      /* sum of HIP-event elapsed times on the launch stream */
    double bytes_per_launch; /* algorithmic bytes of one launch */
} Q3ProfEntry;
/* Event-bracket every kernel launch of subsequent forward() calls (forces the
 * non-graph path while on). */
void q3_prof_enable(Model* m, int on);
void q3_prof_reset(Model* m);
int q3_prof_get(Model* m, Q3ProfEntry* out, int max_entries);
/* Mean duration of the launches of kernel class `name` by the device clock read inside the
 * kernel (first workgroup in .. last workgroup out; GEMV classes only), in microseconds. */
double q3_prof_device_us(Model* m, const char* name);

/* ---- multi-GPU layer pipeline (one process per GPU) --------------------
 * The reference has no multi-device path; this is the layer pipeline
 * BASELINE.json asks for.  Rank r owns a contiguous block of layers (rank 0 also the
 * embedding, the last rank the final norm + classifier); the fp32 residual x[dim]
 * moves from rank r to r+1, and the chosen token id from the last rank back to rank
 * 0, by RCCL send/recv over xGMI.  `world` independent token streams travel around
 * the ring so that every stage works on every tick (stream s, token k is on rank r at
 * tick s + k*world + r); copying fp32 is exact, so each stream's tokens are the ones
 * a single GPU produces. */
#define Q3_PIPE_ID_BYTES 128
/* Rank 0 creates the RCCL unique id; the harness hands the bytes to every rank. */
int q3_pipeline_unique_id(void* id_bytes);
/* Join the communicator; must be called before q3_device_attach(). */
int q3_pipeline_init(int rank, int world, const void* id_bytes);
/* ranks of the RCCL communicator the pipeline runs on (1 = no pipeline) */
int q3_pipeline_size(void);
void q3_pipeline_layers(const ModelParams* p, int rank, int world, int* first, int* count);
/* Which (stream, token index) rank `rank` handles at global tick `tick`; returns 0 when
 * the rank is idle on that tick (pipeline fill / drain). */
int q3_pipeline_schedule(int rank, int world, int nsteps, int tick, int* stream, int* k);
/* Greedy-decode `nsteps` tokens of every stream on the device(s), all streams starting
 * from `first_token` at position pos0.  Asynchronous: returns after enqueueing; follow
 * with q3_device_sync().  With world == 1 this is the on-device greedy loop.
 * Returns the number of ticks on which THIS rank computed: nsteps * world (every rank
 * handles every token of every stream once); it exits with a message on bad arguments,
 * so there is no error value. */
int q3_pipeline_run(Model* m, int first_token, int pos0, int nsteps);
/* The same with only the first `streams` of the `world` streams running (0 or >= world: all).  The tick
 * schedule does not change -- the ticks of a stream that does not run stay idle -- so streams = 1 is ONE
 * token stream travelling through the stages with a hand-off per stage and token (a single
 * stream gains capacity from the pipeline, not speed).  Returns nsteps * streams. */
int q3_pipeline_run_streams(Model* m, int first_token, int pos0, int nsteps, int streams);
/* Tokens stream `stream` chose in the last q3_pipeline_run (valid on the last rank). */
int q3_pipeline_tokens(Model* m, int stream, int* out, int n);
/* Single-GPU self-test of the pipeline code: `world` stages in one process, hand-offs by
 * device copies instead of RCCL.  out_tokens[world][nsteps].  Returns 0 on success. */
int q3_pipeline_selftest(const char* path, int seq_len, int world, int first_token, int pos0,
                         int nsteps, int* out_tokens);
/* ... with only the first `streams` streams running (0 = all): out_tokens[streams][nsteps] */
int q3_pipeline_selftest_streams(const char* path, int seq_len, int world, int streams, int first_token,
                                 int pos0, int nsteps, int* out_tokens);
/* max over ranks (doubles as a barrier) */
double q3_pipeline_allreduce_max(double v);
void q3_pipeline_shutdown(void);

/* diagnostic builds (-DQ3_ATTN_STAMPS, Q3_STAMPS=1): time stamps of the attention kernel */
int q3_debug_stamps(Model* m, unsigned long long* out, int n);
/* Diagnostic: mean microseconds per launch of `iters` back-to-back launches of one GEMV class
 * ("qkv", "wo", "gateup", "down") cycling over layers [l_lo, l_hi). */
double q3_debug_gemv_loop(Model* m, const char* which, int l_lo, int l_hi, int iters);
/* Diagnostic: the same loop through the int8-MFMA GEMM of the prompt path on `ntok` (1..64) activation rows
 * (ntok = 1: the decode GEMV's work on v_mfma_i32_16x16x64_i8, bit-identical results). */
double q3_debug_gemm_loop(Model* m, const char* which, int ntok, int l_lo, int l_hi, int iters);

const char* q3_version(void);

#ifdef __cplusplus
}
#endif
#endif /* Q3_EXT_H */
