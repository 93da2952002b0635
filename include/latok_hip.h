/* latok_hip.h -- C ABI of liblatok_hip.so: latok's character-feature-matrix + split-mask path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference binds this path through the CPython extension module `latok.latok`
 * (reference setup.py:10-18, method table latok/core/src/latok/latok.c:373-378) whose three functions are called by
 * latok/core/default_tokenizer.py:36,123-129,146 and latok/core/latok_utils.py:7,15,24.  Each entry point below names
 * the reference interface it replaces.  Plain pointers and sizes only: no Python.h, no NumPy C-API, no torch types.
 *
 * Conventions
 *   - every function returns LATOK_OK (0) or a negative LATOK_ERR_*; latok_last_error() gives the message
 *     (the reference raises ValueError for bad arguments: latok.c:40-50,151-171,292-312; the Python mirror maps
 *     LATOK_ERR_INVALID -> ValueError and everything else -> RuntimeError).
 *   - the caller owns every buffer (the reference returns freshly allocated NumPy arrays: latok.c:59,174,357).
 *   - `flags & LATOK_DEVICE_PTRS`: all data pointers are device pointers, the call is asynchronous on `stream`
 *     (a hipStream_t, NULL = the library's own stream) and returns as soon as the work is enqueued.  Otherwise they
 *     are host pointers: the library stages through its own device buffers and the call is synchronous.
 *   - a batch is CSR: `cps` = packed UTF-32 code points of all strings, `row_off[n_str + 1]` = start of each string
 *     (row_off[0] == 0, non-decreasing).  Device `cps` pointers must be 16-byte aligned.  Host-pointer calls check
 *     row_off and fail with LATOK_ERR_INVALID; device-resident row offsets cannot be checked without a copy, so a caller
 *     that passes device pointers guarantees them (results are undefined otherwise).
 *   - `total_chars` = row_off[n_str]; the caller normally knows it.  Pass -1 to let the library read it (in device
 *     mode that costs one blocking 8-byte device->host copy).
 *   - contexts, threads and streams: every entry point works on the calling thread's CURRENT CONTEXT (see "contexts"
 *     below; default = the process-wide one latok_init creates).  A context owns one device, one stream, one set of
 *     device workspaces and one lock: calls on the same context are serialised and ordered on the device (a call waits
 *     for the previous call's last kernel before its own first one, whatever stream it uses); calls on DIFFERENT
 *     contexts share nothing and run concurrently -- one host thread + one context per GPU is how a batch is sharded
 *     over the GPUs of a node (SURVEY 8b "Threading", 8e).  The compaction entry points (offsets / spans / features)
 *     return the item total to the host and therefore block even in device mode: everything is enqueued first (the
 *     kernels write the records only if the total fits the caller's capacity) and the call synchronises once.
 *   - small host batches (host pointers, at most LATOK_TILE_CHARS chars and 512 strings -- one string per call is the
 *     reference's own calling pattern, default_tokenizer.py:137-191) are one single-wavefront launch for offsets, spans
 *     and features alike: inputs and outputs pass through pinned memory the kernel reads / writes directly, and the
 *     call returns when it has seen the completion word the kernel stores after its last output (LATOK_SMALL_POLL=0 in
 *     the environment: wait for the stream instead).  Host batches up to 256 K chars / 16 K strings take the same pinned
 *     route with a handful of launches (up to 24 tiles: one launch for the whole mask pipeline; LATOK_ONE_SEGMENT=0: three).
 *     Small host batches of PEP 393 kind 1 / 2 units are widened, and small well-formed UTF-8 batches decoded, by the host
 *     on their way into that pinned area (byte-space results are mapped back to byte positions), so one string per call
 *     costs the same whatever form it arrives in; malformed UTF-8 and larger batches are read by the device as they are.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with LATOK_ERR_HIP.
 */
#ifndef LATOK_HIP_H
#define LATOK_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LATOK_OK 0
#define LATOK_ERR_INVALID (-1)  /* bad argument / shape (reference: ValueError) */
#define LATOK_ERR_HIP (-2)      /* HIP runtime error, or no device */
#define LATOK_ERR_NOT_INIT (-3) /* latok_init() has not been called */
#define LATOK_ERR_NOMEM (-4)

#define LATOK_DEVICE_PTRS 1
/* Compaction entry points (offsets / spans / features): counts_out and the offsets / spans / spans4 records are int32
 * arrays instead of int64 (the parameters keep their int64_t* type: pass the int32 buffer through a cast; capacities stay
 * in ELEMENTS / tokens).  The records are most of the bytes these calls write and send over the bus -- 8 B per boundary,
 * 16 B per token -- so the 32-bit form halves that.  Every value is relative to its own string, so it fits unless a
 * single string has 2^31 chars or more: then the call fails with LATOK_ERR_INVALID and the 64-bit form has to be used. */
#define LATOK_OUT_INT32 2

#define LATOK_FEATURE_COUNT 25 /* reference latok/core/offsets.py:49 */
#define LATOK_TILE_CHARS 4096  /* chars per wavefront tile (64 lanes x 64-bit words) */

/* ---- lifecycle ------------------------------------------------------------------------------------------------ */
int latok_device_count(void);          /* number of HIP devices, 0 when none (never fails) */
int latok_init(int device);            /* create the DEFAULT context on `device`: Unicode tables, stream, workspaces */
int latok_shutdown(void);              /* destroy the default context */
const char* latok_last_error(void);    /* message of the last failure on this thread */
const char* latok_version(void);

/* ---- contexts: one per GPU (or several per GPU) in one process -------------------------------------------------------
 * The reference holds the GIL for a whole call and has no state (latok.c:373-378: three pure functions), so it is
 * re-entrant but never concurrent.  Here the state a call needs (device, stream, tables, workspaces, run-time rule
 * tables) lives in a context.  latok_ctx_create binds a new context to `device`; latok_ctx_set_current makes it the
 * calling THREAD's current context (NULL = back to the default one) -- the model of hipSetDevice -- and every entry point
 * of this header then runs on it: its device, its stream, its rule tables (latok_set_rules is per context).  The
 * caller's current HIP device is never changed by a call.  A context must not be destroyed while another thread still
 * has it current.  Device memory from latok_dev_alloc belongs to the device of the context that was current. */
typedef struct latok_ctx latok_ctx;
int latok_ctx_create(int device, latok_ctx** ctx_out);
int latok_ctx_destroy(latok_ctx* ctx);
int latok_ctx_set_current(latok_ctx* ctx);   /* NULL = the default context */
latok_ctx* latok_ctx_get_current(void);      /* NULL when the thread runs on the default context */
int latok_ctx_device(latok_ctx* ctx);        /* device of a context (NULL = the default one), -1 when not initialised */

/* Grow the library-owned workspace (tile summaries, segment aggregates) for batches of up to
 * `max_chars` code points / `max_strings` strings, so that later calls allocate nothing. */
int latok_reserve(int64_t max_chars, int64_t max_strings);

/* ---- the fused hot path ------------------------------------------------------------------------------------------
 * Replaces, for a whole batch at once, the reference call chain of default_tokenizer.py:146-148:
 *   _gen_parse_matrix (latok.c:31-138) -> gen_split_mask (default_tokenizer.py:113-134: 3x _combine_matrix_rows
 *   latok.c:275-370 + gen_block_mask latok.c:140-258) -> nonzero-ness of the result.
 * mask_bits_out: uint64[ceil(total_chars / 64)], bit (i & 63) of word (i >> 6) = 1 iff packed char i is a token
 * boundary (splits[i] != 0).  Every string's first char is a boundary (default_tokenizer.py:132); empty strings
 * contribute nothing (the reference raises IndexError for '' -- documented deviation of batch mode). */
int latok_split_mask_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                           uint64_t* mask_bits_out, int flags, void* stream);

/* Same pipeline, but writes the reference's split VALUES (0..5, the int8 vector returned by gen_split_mask,
 * default_tokenizer.py:121-134) as uint8[total_chars].  Parity/debug form: 1 byte per char instead of 1 bit. */
int latok_split_values_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                             uint8_t* values_out, int flags, void* stream);

/* Boundary offsets (replaces np.nonzero(splits)[0], default_tokenizer.py:148) for every string of the batch:
 * counts_out[n_str] = number of boundaries of each string; offsets_out[0..sum(counts)) = the offsets of string 0,
 * then string 1, ... each relative to its own string start (int64, ascending, first one always 0).
 * offsets_cap = capacity of offsets_out in elements; *n_offsets_out = total number written (host pointer, always).
 * Returns LATOK_ERR_INVALID if offsets_cap is too small (n_offsets_out still holds the needed size).
 * Synchronous in both pointer modes (the total is returned to the host). */
int latok_split_offsets_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                              int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap, int64_t* n_offsets_out,
                              int flags, void* stream);

/* Token spans: everything tokenize() does after np.nonzero (reference default_tokenizer.py:149-158), on the device:
 * consecutive boundaries delimit a token, leading/trailing SPACE-class chars are stripped (str.strip()), whitespace-only
 * tokens are dropped.  counts_out[n_str] = tokens per string; spans_out[2*k], spans_out[2*k+1] = [start, end) of token
 * k relative to its string start (tokens of string 0 first).  spans_cap = capacity in tokens; *n_tokens_out = total
 * (host pointer).  LATOK_ERR_INVALID when spans_cap is too small (n_tokens_out still holds the needed count).
 * Synchronous in both pointer modes. */
int latok_token_spans_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                            int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                            int flags, void* stream);

/* ---- UTF-8 ingest (one step before the path: the reference reads CPython's PEP-393 buffer, latok.c:53-55,79) ---------
 * A batch can also be handed over as UTF-8: `utf8` = packed bytes of all strings, `byte_off[n_str + 1]` = byte offset
 * of each string (byte_off[0] == 0).  The library decodes on the device (one code point per lead byte; input must be
 * valid UTF-8, "surrogatepass" forms decode as they are, truncated sequences give U+FFFD) and runs the same pipeline.
 * All results are in CODE-POINT units, exactly what the reference would report for the decoded str.  With host
 * pointers this moves 1 byte per ASCII char over PCIe instead of 4. */
int latok_utf8_decode_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                            uint32_t* cps_out, int64_t cps_cap, int64_t* cp_row_off_out, int64_t* total_cps_out, int flags,
                            void* stream);
/* boundary bitmask over the DECODED code points (bit i = code point i of the packed batch) + the code-point row offsets */
int latok_split_mask_utf8_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                uint64_t* mask_bits_out, int64_t mask_cap_words, int64_t* cp_row_off_out,
                                int64_t* total_cps_out, int flags, void* stream);
int latok_split_offsets_utf8_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                   int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap,
                                   int64_t* n_offsets_out, int flags, void* stream);
int latok_token_spans_utf8_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                 int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                                 int flags, void* stream);

/* ---- UTF-8 in BYTE space (fused ingest) -----------------------------------------------------------------------------
 * Same tokenization, but nothing is decoded to UTF-32: the tile kernel reads the UTF-8 bytes themselves (1 byte per
 * ASCII char from HBM instead of 4) and every position it reports is a BYTE position in the caller's buffer:
 *   mask     bit i = byte i of the packed buffer; set at the LEAD byte of every char the reference marks as a boundary
 *            (np.nonzero(gen_split_mask(...)), default_tokenizer.py:148), mapped from code-point to byte positions
 *   offsets  byte offsets relative to the start of each string
 *   spans    [start, end) byte ranges of the stripped, non-empty tokens: utf8[byte_off[s] + start : byte_off[s] + end]
 *            is the UTF-8 encoding of the token the reference yields
 * Input must be valid UTF-8 ("surrogatepass" forms are accepted as they decode); a truncated sequence counts as
 * U+FFFD, stray continuation bytes belong to no char.  With LATOK_DEVICE_PTRS the byte buffer must be 16-byte aligned.
 * Run-time rule tables (latok_set_rules) apply in byte space too (evaluated by the byte-space tile kernel itself). */
int latok_split_mask_utf8_bytes_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                      uint64_t* mask_bits_out, int flags, void* stream);
int latok_split_offsets_utf8_bytes_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                         int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap,
                                         int64_t* n_offsets_out, int flags, void* stream);
int latok_token_spans_utf8_bytes_batch(const uint8_t* utf8, const int64_t* byte_off, int64_t n_str, int64_t total_bytes,
                                       int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                                       int flags, void* stream);

/* Token feature vectors: reference featurize() (default_tokenizer.py:163-191) for a whole batch without the n x 25
 * matrix.  Per kept token k: spans4_out[4k..4k+3] = {raw_start, raw_end, strip_start, strip_end} (LaToken.start_idx /
 * end_idx are the raw span, LaToken.text is text[strip_start:strip_end]); features_out[25k..25k+24] = sum of the 25
 * feature columns over the raw span in uint8 wrap-around arithmetic (latok.c:342-354).  cap in tokens. */
int latok_token_features_batch(const uint32_t* cps, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                               int64_t* counts_out, int64_t* spans4_out, int8_t* features_out, int64_t cap,
                               int64_t* n_tokens_out, int flags, void* stream);

/* ---- PEP 393 buffers: the reference's own input format ------------------------------------------------------------
 * The reference reads a str through PyUnicode_KIND / PyUnicode_DATA (latok.c:53-55,79): fixed-width code units of
 * kind = 1 (Latin-1), 2 (UCS-2) or 4 (UCS-4) bytes.  These entry points take that buffer as it is: `units` = the
 * packed units of all strings, row_off / total_chars / every result in units = chars (CPython stores text with astral
 * chars as kind 4, so a kind-2 unit is always a whole code point; lone surrogates are classified as the code points
 * they are).  A caller that holds Python strings never widens them to UTF-32, and a Latin-1 / UCS-2 batch costs 1 / 2
 * bytes per char on the bus and in HBM: the tile kernel reads the narrow units itself (mask, offsets, spans), under
 * run-time rule tables as well; only featurize widens them once on the device (it re-reads the code points).  Results
 * are identical to the UTF-32 entry points on the widened text.  kind = 4 forwards to those.  With LATOK_DEVICE_PTRS `units` must be 16-byte aligned. */
int latok_split_mask_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                uint64_t* mask_bits_out, int flags, void* stream);
int latok_split_offsets_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                   int64_t* counts_out, int64_t* offsets_out, int64_t offsets_cap, int64_t* n_offsets_out,
                                   int flags, void* stream);
int latok_token_spans_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                 int64_t* counts_out, int64_t* spans_out, int64_t spans_cap, int64_t* n_tokens_out,
                                 int flags, void* stream);
int latok_token_features_kind_batch(const void* units, int kind, const int64_t* row_off, int64_t n_str, int64_t total_chars,
                                    int64_t* counts_out, int64_t* spans4_out, int8_t* features_out, int64_t cap,
                                    int64_t* n_tokens_out, int flags, void* stream);

/* ---- the reference's three native functions, one string at a time (compat surface) ---------------------------- */
/* _gen_parse_matrix (latok.c:31-138): n code points -> int8[n][25], C-contiguous. */
int latok_parse_matrix(const uint32_t* cps, int64_t n, int8_t* matrix_out, int flags, void* stream);

/* _combine_matrix_rows (latok.c:275-370): m is a 2-D byte matrix addressed m[r*stride_r + c*stride_c] with `rows`
 * rows and `cols` columns; idx is int8, idx_ndim 2 (irows x icols, "sum of products", -1 skipped) or 1 (icols row
 * ids, "sum").  out = int8[cols].  uint8 wrap-around arithmetic like the reference.  The matrix is copied densely
 * (rows x cols) before upload when given as host pointers; in device mode strides are honoured as given. */
int latok_combine_matrix_rows(const int8_t* m, int64_t rows, int64_t cols, int64_t stride_r, int64_t stride_c,
                              const int8_t* idx, int idx_ndim, int irows, int icols, int8_t* out, int flags,
                              void* stream);

/* _gen_block_mask (latok.c:140-258): a1 ("starts") and a2 ("spaces") are int8[n], non-zero = set. out = int8[n]. */
int latok_block_mask(const int8_t* a1, const int8_t* a2, int64_t n, int8_t* out, int flags, void* stream);

/* ---- device memory / stream helpers (so hosts need no other GPU runtime binding) ------------------------------ */
void* latok_dev_alloc(size_t bytes);   /* NULL on failure */
int latok_dev_free(void* p);
/* Pinned (page-locked) host memory: host-pointer batches handed over in such buffers move over the bus at full speed and
 * asynchronously -- the chunked pipeline of the large-batch compaction calls then keeps both copy directions busy at once. */
void* latok_host_alloc(size_t bytes);  /* NULL on failure */
int latok_host_free(void* p);
int latok_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int latok_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int latok_memset_dev(void* dst_dev, int value, size_t bytes);
int latok_sync(void);                  /* wait for the library stream */
int latok_device_props(int* n_cu, int64_t* hbm_bytes, char* name_out, int name_cap);

/* ---- synthetic corpora (counter-based, identical on host and device) ------------------------------ */
#define LATOK_CORPUS_ASCII 0
#define LATOK_CORPUS_UNICODE 1
/* row_off_out[n_str + 1] (host): lengths uniform in [len_lo, len_hi] for string ids sid0 .. sid0+n_str-1 */
int latok_corpus_offsets(uint64_t seed, uint64_t sid0, int64_t n_str, int64_t len_lo, int64_t len_hi,
                         int64_t* row_off_out);
/* fill code points; host form (pure CPU, no device needed) and device form (one thread per string) */
int latok_corpus_fill_host(uint64_t seed, int model, uint64_t sid0, int64_t n_str, const int64_t* row_off,
                           uint32_t* cps_out);
int latok_corpus_fill_device(uint64_t seed, int model, uint64_t sid0, int64_t n_str, const int64_t* row_off_dev,
                             uint32_t* cps_out_dev, void* stream);
/* total UTF-8 encoded size of n code points (device pointers when LATOK_DEVICE_PTRS); result to a host int64 */
int latok_utf8_bytes(const uint32_t* cps, int64_t n, int64_t* bytes_out, int flags);

/* ---- runtime rule tables ------------------------------------------------------------------------------------------
 * The reference's extension point (latok/core/default_tokenizer.py:9-30,108-110): other C_SPLIT / C_MASK / C_SYM
 * matrices built with build_combo_matrix (latok/core/latok_utils.py:27-56) over the 25 feature columns
 * (latok/core/offsets.py:24-49), combined as gen_split_mask does (default_tokenizer.py:113-134):
 *     splits = combine(C_SPLIT) * block_mask(combine(C_MASK), SPACE) + combine(C_SYM);  splits[0] = 1
 * with combine = _combine_matrix_rows (latok.c:275-370).  After latok_set_rules EVERY batch entry point evaluates the
 * caller's tables inside the fused kernel, for every input form -- UTF-32, UTF-8 in byte space, UTF-8 in code-point
 * units, PEP 393 kind 1 / 2 units (each has its own tile-kernel instantiation; nothing is widened or decoded first) --
 * and every output: bitmask, offsets, token spans, featurize, and latok_split_values_batch, which then returns what
 * gen_split_mask returns for those tables: (number of C_SPLIT rows that hold) * mask + (number of C_SYM rows that hold),
 * 1 at a string start.  Boundaries and values are bit-exact with the reference recipe run on the same tables.  Each
 * table is a row-major int8 [rows x cols] matrix of column ids, -1 padding short rows; limits: <= 32 rows per table (the
 * rows travel in the kernel arguments; the reference's own tables have 5 / 4 / 1), ids 0..24, a row must not START
 * with -1 (the reference would reuse the previous row's product there).  rows = 0 gives the all-zero vector.  State of
 * the current context. */
int latok_set_rules(const int8_t* c_split, int split_rows, int split_cols, const int8_t* c_mask, int mask_rows,
                    int mask_cols, const int8_t* c_sym, int sym_rows, int sym_cols);
int latok_reset_rules(void);   /* back to the built-in default_tokenizer.py tables */
int latok_rules_active(void);  /* 1 while custom tables are installed */

/* ---- batch flow: many device-resident batches through one context, overlapped -------------------------------------------
 * The reference tokenizes one string after another (default_tokenizer.py:137-160: every call is independent of the one
 * before).  Here a batch costs three dependent launches (per-tile string index, tiles, resolve); only the tile kernel needs
 * the whole GPU.  A flow keeps up to TWO batches in flight on the current context -- every batch in flight on a stream and
 * a workspace set of its own (a "slot"; submissions take the slots in turn), with no dependency between the slots -- so that
 * the two small launches of one batch run beside the tile kernel of the other; a flow batch's tile kernel is planned for 7/8 of
 * the CUs, so consecutive tile kernels overlap their start-up and ragged end as well (C2: 0.108 -> 0.087 ms per batch).
 *   latok_flow_split_mask: enqueue latok_split_mask_batch(LATOK_DEVICE_PTRS) of one batch and return.  The inputs must be
 *     complete in device memory when the call is made (they are NOT ordered behind work on any caller stream), and must stay
 *     untouched by the caller until latok_flow_wait.  total_chars < 0: read from row_off (one small synchronous copy).
 *     Results are bit-identical to latok_split_mask_batch.
 *   latok_flow_wait: block until every batch submitted on the current context is complete (latok_sync does the same).  The
 *     streams are polled for up to 2 ms before the call sleeps on them (a sleeping wait returns ~15 us late).
 * Ordering between batches of one flow.  Every call is independent, as the reference's calls are -- batches that touch
 * disjoint memory overlap freely.  The library tracks the byte RANGE of every buffer a batch in flight reads (units, row
 * offsets) or writes (mask; for the compaction calls: records, counts, result words, feature sums) until the flow is next
 * idle.  A new batch that writes any byte a batch still in flight reads or writes, or reads one it writes -- whole buffer or
 * partial overlap, however many other batches were submitted in between -- is ordered behind that batch (it is enqueued on that
 * batch's slot; if batches on both slots are in its way the call first waits for the flow to drain).  So reusing an output
 * buffer is always correct, it just does not overlap; callers that want the overlap alternate their buffers.
 * The blocking entry points may be called on the same context while a flow is in flight (they use the context's own stream
 * and workspace; their buffers are NOT tracked against the flow's).  A batch larger than any its slot has seen grows the
 * slot's workspace, which first waits for the flow to drain. */
int latok_flow_split_mask(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                          uint64_t* mask_dev);
/* The same for the other input forms of the path: PEP 393 units (kind 1 / 2 / 4 as latok_split_mask_kind_batch; positions are
 * chars) and UTF-8 in byte space (as latok_split_mask_utf8_bytes_batch; positions are bytes).  Batches of different forms may
 * follow each other in one flow. */
int latok_flow_split_mask_kind(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                               uint64_t* mask_dev);
int latok_flow_split_mask_utf8_bytes(const uint8_t* utf8_dev, const int64_t* byte_off_dev, int64_t n_str, int64_t total_bytes,
                                     uint64_t* mask_dev);
/* Boundary offsets / token spans of a batch through the flow: what latok_split_offsets_batch / latok_token_spans_batch (and
 * their _kind / _utf8_bytes forms) compute, enqueued without waiting for the item total.  kind: 4 = UTF-32 code points, 1 / 2 =
 * PEP 393 units, 0 = UTF-8 bytes in byte space (row_off = byte offsets, positions are bytes).  Every pointer is a device
 * address (LATOK_DEVICE_PTRS is implied); flags: LATOK_OUT_INT32 for int32 counts / records.
 *   counts_dev[n_str], offsets_dev[offsets_cap] (spans_dev[2 * spans_cap]): as in the blocking calls.
 *   result_dev: int64[2] in memory the DEVICE can write and the caller can read after latok_flow_wait (latok_dev_alloc +
 *     latok_memcpy_d2h, or latok_host_alloc for a direct read): result[0] = number of items of the batch, result[1] = 0, or
 *     nonzero when the batch could not be reported (low half: a string of >= 2^31 chars under LATOK_OUT_INT32; high half:
 *     internal scan error, the call is safe to repeat).  When result[0] exceeds the capacity nothing was written to the
 *     records (counts are valid): resubmit with a larger buffer -- the capacity protocol of the blocking calls, read late.
 * Every output (records, counts, result words) takes part in the ordering rule above. */
int latok_flow_split_offsets(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_units,
                             void* counts_dev, void* offsets_dev, int64_t offsets_cap, int64_t* result_dev, int flags);
int latok_flow_token_spans(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_units,
                           void* counts_dev, void* spans_dev, int64_t spans_cap, int64_t* result_dev, int flags);
/* featurize through the flow: latok_token_features_batch / _kind_batch (kind 4 / 1 / 2) with the same result words */
int latok_flow_token_features(const void* units_dev, int kind, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                              void* counts_dev, void* spans4_dev, int8_t* features_dev, int64_t cap, int64_t* result_dev, int flags);
int latok_flow_wait(void);

/* ---- measurement ----------------------------------------------------------------------------------------------- */
/* Run latok_split_mask_batch on device-resident data: `warmup` untimed passes, then
 *   ms_total_out  = elapsed ms of `iters` whole-pipeline passes (tile index, tiles, resolve) between ONE pair of HIP events
 *                   on the stream the kernels run on;
 *   ms_tiles_out  = elapsed ms of `iters` further launches of the dominant kernel alone (k_tiles_main, back to back,
 *                   again one event pair: an event pair per launch charges each interval with ~6 us of marker dispatch);
 *   n_fix_tiles_out = tiles recomputed by the resolve stage in the last pass.
 * Any may be NULL (a NULL output skips its passes), so warm-up-only and kernel-only calls are possible. */
int latok_bench_split_mask(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                           uint64_t* mask_dev, int warmup, int iters, float* ms_total_out, float* ms_tiles_out,
                           int64_t* n_fix_tiles_out);

/* Streaming-read ceiling of this GPU: time `iters` launches of a kernel that only reads `bytes` (a multiple of 16 KiB
 * is used; device pointer, 16-byte aligned) with the tile kernel's load pattern.  ms_out = elapsed ms of the `iters`
 * launches after `warmup` untimed ones.  Measurement aid for SURVEY.md 8(d) ("vs. a measured streaming-read kernel"). */
int latok_bench_stream_read(const void* buf_dev, int64_t bytes, int warmup, int iters, float* ms_out);

/* Several contexts in one process, timed as ONE job (one host thread + one context per GPU, no collective).
 * A gate is a rendezvous of `parties` host threads that lives outside any context (no device needed):
 * latok_gate_wait returns on every thread once all of them have arrived (it spins inside the library, so a ctypes
 * caller has released the GIL) and fails with LATOK_ERR_INVALID after `timeout_s` seconds without the full set.  A gate
 * can be passed any number of times.  latok_gate_break makes every present and future wait fail at once (a party that
 * cannot reach the gate calls it so that the others do not sit out the timeout).
 * latok_bench_split_mask_gated = the timed region of the whole-job measurement on the CURRENT context:
 *   wait at `gate` (NULL: no wait) -> host clock t0 -> HIP event -> `iters` whole-pipeline passes -> HIP event ->
 *   stream synchronise -> host clock t1 -> wait at `gate` again (so that no thread starts anything else on the node while
 *   another is still inside its region).
 * ms_events_out = the event pair; t0_ns_out / t1_ns_out = the host's monotonic clock (one clock for every thread of the
 * process), so the job took max(t1) - min(t0) over the contexts. */
typedef struct latok_gate latok_gate;
int latok_gate_create(int parties, latok_gate** gate_out);
int latok_gate_destroy(latok_gate* gate);
/* The same rendezvous for one PROCESS per GPU: the gate lives in POSIX shared memory under `name` ("/something").  One
 * process creates it, the others attach; every process detaches when done and one of them unlinks the name.  The host's
 * monotonic clock is one clock for all processes of the machine, so t0 / t1 of latok_bench_split_mask_gated compare. */
int latok_gate_create_shared(const char* name, int parties, latok_gate** gate_out);
int latok_gate_attach_shared(const char* name, latok_gate** gate_out);
int latok_gate_detach_shared(latok_gate* gate);
int latok_gate_unlink_shared(const char* name);
int latok_gate_wait(latok_gate* gate, double timeout_s);
int latok_gate_break(latok_gate* gate);
int latok_bench_split_mask_gated(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                                 uint64_t* mask_dev, int iters, latok_gate* gate, float* ms_events_out,
                                 int64_t* t0_ns_out, int64_t* t1_ns_out);
/* The same timed region through the batch flow: `iters` latok_flow_split_mask submissions of the batch, writing mask_a_dev
 * and mask_b_dev alternately (two buffers of ceil(total_chars / 64) words), then latok_flow_wait -- every pass does all
 * three launches and completes inside the region.  ms_events_out: from an event in front of the first string-index launch
 * to one behind the last resolve launch. */
int latok_bench_split_mask_flow_gated(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                                      uint64_t* mask_a_dev, uint64_t* mask_b_dev, int iters, latok_gate* gate,
                                      float* ms_events_out, int64_t* t0_ns_out, int64_t* t1_ns_out);

/* A second, equal copy of the batch at another address for the two flow measurements (this one and latok_bench_tiles_flow): the
 * odd steps then read cps_b_dev / row_off_b_dev instead, so that the two batches in flight share no input lines in L2 / MALL -- as
 * two different batches of a real flow would not.  NULL, NULL: back to one shared input.  State of the current context. */
int latok_bench_set_second_input(const uint32_t* cps_b_dev, const int64_t* row_off_b_dev);

/* The dominant kernel alone in the flow's launch scheme: `iters` launches of k_tiles_main (planned for 7/8 of the CUs, as every
 * batch of a flow is) alternating between the two slot streams, nothing else launched; ms_out = host wall time from the first
 * launch to the end of the last (streams polled).  The launches overlap, so ms_out / iters is the kernel's average cost per
 * launch in the flow, not the duration of one launch. */
int latok_bench_tiles_flow(const uint32_t* cps_dev, const int64_t* row_off_dev, int64_t n_str, int64_t total_chars,
                           uint64_t* mask_a_dev, uint64_t* mask_b_dev, int iters, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif
