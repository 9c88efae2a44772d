/*
 * lzfear_hip.h — C ABI of the MI355X (gfx950) LZ4 raw-block codec that drops in under
 * lz-fear's `raw` module.
 *
 * This is the boundary a Rust `-sys` crate would bind (INTEGRATION.md shows the `extern "C"`
 * block and the safe wrappers with the reference's own signatures).  Plain pointers and sizes
 * only; no C++ or torch types.  Each entry point cites the reference interface it replaces
 * (file:line in the lz-fear 0.2.0 tree).
 *
 * The reference calls its codec once per frame block (src/framed/compress.rs:243,
 * src/framed/decompress.rs:248).  The GPU boundary is the same call *batched over jobs*: one
 * job = one `compress2` / `decompress_raw` invocation, one wavefront (compress) or one
 * workgroup (decompress) each, many jobs per launch.
 *
 * Memory: every pointer in a job is a DEVICE pointer (HBM) unless the function name ends in
 * `_host`.  The library never retains pointers past a call and owns no global state besides
 * the staging memory of the `_host` helpers and the frame layer (one pinned slab and device
 * scratch, kept between calls; lzf_frame_release_scratch() of lzfear_frame.h frees them) and ONE process-wide setting: the batch
 * calls take their scratch from the device's default stream-ordered memory pool and raise that pool's release threshold
 * (hipMemPoolAttrReleaseThreshold) to "keep", once per device — without it every call would go back to the device allocator
 * and wait for whatever is running.  There is NO CPU fallback: every entry point
 * returns LZF_E_NO_DEVICE when no HIP device is usable.
 */
#ifndef LZFEAR_HIP_H
#define LZFEAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZFEAR_ABI_VERSION 2

/* ---- per-job status (lzf_job_result.status) --------------------------------------------- */
enum {
    LZF_OK = 0,
    /* raw::DecodeError — src/raw/decompress.rs:8-17 */
    LZF_UNEXPECTED_END = 1,
    LZF_MEMORY_LIMIT_EXCEEDED = 2,
    LZF_ZERO_DEDUP_OFFSET = 3,
    LZF_INVALID_DEDUP_OFFSET = 4,
    /* compress2's writer refused a write: io::ErrorKind::ConnectionAborted from
     * NoPartialWrites — src/framed/compress.rs:294-314, matched at :250-255 */
    LZF_OUTPUT_FULL = 5,
    /* the reference would panic: assert at src/raw/compress/mod.rs:167, expect at :67/:92 */
    LZF_CONTRACT = 6,
    /* decode output buffer smaller than what the reference's Vec would have grown to
     * (literals are not limit-checked, src/raw/decompress.rs:63-67): give the job
     * out_cap >= output_limit + input_len and this never happens */
    LZF_OUT_CAPACITY = 7
};

/* ---- library-level return codes (negative) ---------------------------------------------- */
enum {
    LZF_E_NO_DEVICE = -1,     /* no HIP device / HIP runtime error at init */
    LZF_E_HIP = -2,           /* a HIP call failed; lzf_last_error() has the text */
    LZF_E_INVALID = -3        /* bad argument (NULL jobs with n_jobs > 0, ...) */
};

/* ---- encoder tables: src/raw/compress/mod.rs:27-36 (U32Table), :78-87 (U16Table) --------- */
#define LZF_TABLE_U32 0       /* 4096 x u32, payload limit u32::MAX (:75)  — what framed uses (:202) */
#define LZF_TABLE_U16 1       /* 8192 x u16, payload limit 65535   (:100) — raw API only        */

typedef struct lzf_u32_table { uint32_t dict[4096]; uint64_t offset; } lzf_u32_table;
typedef struct lzf_u16_table { uint16_t dict[8192]; uint64_t offset; } lzf_u16_table;

/* ---- jobs ------------------------------------------------------------------------------- */

/* One `compress2(input, cursor, &mut table, writer)` call — src/raw/compress/mod.rs:165-166.
 *   input[0..cursor) is addressable history (dictionary / previous-block window),
 *   input[cursor..input_len) is the payload.  The writer is a bounded sink of `out_cap` bytes
 *   with NoPartialWrites semantics (the frame layer passes out_cap = payload length,
 *   src/framed/compress.rs:242).
 *   table == NULL  <=>  `&mut T::default()` thrown away afterwards (independent blocks without
 *   a dictionary).  Otherwise it points at an lzf_u32_table / lzf_u16_table in device memory
 *   which is read at start and written back at the end (mutations survive LZF_OUTPUT_FULL,
 *   exactly like the reference's `&mut table`). */
typedef struct lzf_compress_job {
    const uint8_t* input;
    uint64_t input_len;
    uint64_t cursor;
    uint8_t* out;
    uint64_t out_cap;
    void* table;
    uint32_t table_kind;          /* LZF_TABLE_U32 | LZF_TABLE_U16 */
    uint32_t flags;               /* LZF_CJOB_* */
} lzf_compress_job;

#define LZF_CJOB_TABLE_READONLY 1u   /* do not write the table back: `template_table.clone()`
                                        per independent block, src/framed/compress.rs:220,270 */

/* One `decompress_raw(input, prefix, &mut output, output_limit)` call —
 * src/raw/decompress.rs:58-59.  `out[0..out_existing_len)` is the Vec's content on entry
 * (addressable history); decoded bytes are appended after it.  out_cap is the room the Vec may
 * grow to. */
typedef struct lzf_decompress_job {
    const uint8_t* input;
    uint64_t input_len;
    const uint8_t* prefix;
    uint64_t prefix_len;
    uint8_t* out;
    uint64_t out_existing_len;
    uint64_t out_cap;
    uint64_t output_limit;
} lzf_decompress_job;

typedef struct lzf_job_result {
    uint64_t out_len;             /* compress: bytes written; decompress: output.len() (incl. existing) */
    int32_t status;               /* LZF_OK ... LZF_OUT_CAPACITY; out_len is unspecified on error */
    uint32_t reserved;            /* diagnostic only: kilo-cycles the job's wavefront ran */
} lzf_job_result;

/* ---- library ---------------------------------------------------------------------------- */
int lzf_abi_version(void);
const char* lzf_last_error(void);
/* Number of usable HIP devices (>= 1) or LZF_E_NO_DEVICE. */
int lzf_device_count(void);

/* Batched raw::compress2 — src/raw/compress/mod.rs:165-238 for every job.
 * d_jobs / d_results are device arrays of n_jobs entries.  Asynchronous on `hip_stream`
 * (a hipStream_t; NULL = the legacy default stream) of the CURRENT device.
 * `table_kinds` says which table types occur in the batch (the job array lives in HBM, the
 * host cannot look): LZF_KINDS_U32, LZF_KINDS_U16 or both or'ed; 0 means "either".
 * Batches larger than the device holds at once are executed longest job first (an internal launch order: results[i]
 * always belongs to jobs[i]). */
#define LZF_KINDS_U32 1u
#define LZF_KINDS_U16 2u
/* optional promise, or'ed in: every U32 job of the batch has table == NULL or a LZF_CJOB_TABLE_READONLY table whose
 * offset is 0, and cursor <= input_len < 2 GiB (the independent-block jobs of src/framed/compress.rs:265-270).  Saves the
 * launch of the general kernel; a job that breaks the promise reports LZF_CONTRACT. */
#define LZF_KINDS_U32_FRESH_ONLY 4u
int lzf_compress_batch(const lzf_compress_job* d_jobs, lzf_job_result* d_results,
                       uint32_t n_jobs, uint32_t table_kinds, void* hip_stream);

/* Batched raw::decompress_raw — src/raw/decompress.rs:58-138 for every job. */
int lzf_decompress_batch(const lzf_decompress_job* d_jobs, lzf_job_result* d_results,
                         uint32_t n_jobs, void* hip_stream);
/* The same with what the caller knows about the batch: an upper bound of the jobs' input_len (the job array lives in HBM, the
 * library cannot look; ~0 = unknown, which is what lzf_decompress_batch passes).  Batches of up to four blocks per CU go through the
 * segmented pipeline (a block decoded by many wavefronts; blocks of 64 KiB .. 4 MiB + 32 KiB of input without prefix / existing
 * output), whose stream-ordered scratch — bit maps, tile sums and a 16-byte record per sequence, up to ~0.15 + 1.8 bytes per byte of
 * `max_input_len` and job, from the device's default memory pool, freed in stream order — is sized by this bound; with a bound below
 * 64 KiB the pipeline is skipped.  Results are identical either way. */
int lzf_decompress_batch_sized(const lzf_decompress_job* d_jobs, lzf_job_result* d_results,
                               uint32_t n_jobs, uint64_t max_input_len, void* hip_stream);
/* Diagnostic: the kernels the calling thread's last lzf_decompress_batch launched (the batch size picks them: the
 * segmented pipeline — one block decoded by many wavefronts — up to four blocks per CU, one workgroup per block beyond). */
const char* lzf_last_decompress_launch(void);
/* The same for the calling thread's last lzf_compress_batch: "lzf_compress_team_kernel" (the latency class: no more jobs than
 * compute units, a team of three wavefronts and a CU's LDS per block) or "lzf_compress_compact_kernel" (one wavefront per
 * block, 18 per CU), plus the general kernels that ran beside it for U16 / writable-table jobs. */
const char* lzf_last_compress_launch(void);

/* EncoderTable helpers on device tables.
 * lzf_table_seed_from_dictionary: the template-table loop of src/framed/compress.rs:202-211
 *   (`for window in dict.windows(8).step_by(3) { template_table.replace(dict, offset) }`).
 * lzf_table_offset: EncoderTable::offset — src/raw/compress/mod.rs:72-74 / :97-99. */
int lzf_table_seed_from_dictionary(lzf_u32_table* d_table, const uint8_t* d_dict,
                                   uint64_t dict_len, void* hip_stream);
int lzf_table_offset(void* d_table, uint32_t table_kind, uint64_t add, void* hip_stream);
/* The same for n tables in one launch (d_tables[i]->offset += d_adds[i]): the `table.offset(forget)` of every
 * linked-block stream of a batch between two blocks, src/framed/compress.rs:271-275. */
int lzf_table_offset_batch(void* const* d_tables, const uint64_t* d_adds, uint32_t n, uint32_t table_kind,
                           void* hip_stream);

/* Linked-block streams decoded without a host round trip per block (src/framed/decompress.rs:238-269 for many
 * streams at once).  A stream's output is one contiguous device buffer (history = everything decoded so far, which
 * covers the reference's 64 KiB carry-over window); before step k this call finishes step k-1 and prepares step k of
 * every stream:
 *   - prev_job[i] (index into d_jobs / d_results, or UINT32_MAX): if that job failed the stream is dead (its later
 *     jobs are emptied), else the stream's length becomes its out_len;
 *   - job[i] (or UINT32_MAX): d_jobs[job[i]].out_existing_len = length, out_cap = length + block_maxsize + input_len,
 *     output_limit = length + block_maxsize;
 *   - stored_len[i] > 0: a stored block — stored_src[i] is copied to the end of the stream's output instead.
 * d_state[i] = {length, dead flag} persists across the calls of one chain. */
typedef struct lzf_chain_state { uint64_t length; uint32_t dead; uint32_t reserved; } lzf_chain_state;
typedef struct lzf_chain_step {
    uint32_t prev_job;              /* job of the previous step, UINT32_MAX if none / stored */
    uint32_t job;                   /* job of this step, UINT32_MAX if none / stored */
    uint64_t stored_len;            /* > 0: this step is a stored block */
    const uint8_t* stored_src;
    uint8_t* out;                   /* the stream's output buffer */
    uint64_t block_maxsize;
} lzf_chain_step;
int lzf_chain_decompress_step(const lzf_chain_step* d_steps, lzf_chain_state* d_state, uint32_t n_streams,
                              lzf_decompress_job* d_jobs, const lzf_job_result* d_results, void* hip_stream);

/* Batched XXH32 (seed 0) of n device buffers: the per-block checksums of
 * src/framed/compress.rs:259-263 and src/framed/decompress.rs:228-235. */
int lzf_xxh32_batch(const uint8_t* const* d_ptrs, const uint64_t* d_lens, uint32_t* d_out,
                    uint32_t n, void* hip_stream);

/* Stored blocks: copies n device byte ranges (d_src[i] -> d_dst[i], d_len[i] bytes, each at most max_len) in one
 * launch — the raw-block moves of src/framed/compress.rs:250-255 and src/framed/decompress.rs:250 without going
 * through the host.  Ranges must not overlap each other. */
int lzf_copy_ranges(const uint8_t* const* d_src, uint8_t* const* d_dst, const uint64_t* d_len, uint32_t n,
                    uint64_t max_len, void* hip_stream);

/* ---- host-buffer convenience (synchronous; stages through device scratch) ---------------
 * Same semantics as the batch calls but every pointer in the jobs is a HOST pointer and
 * `results` is a host array.  `table` pointers are host lzf_*_table structs, updated in
 * place.  Used by the frame layer and by callers that have not moved their data to HBM. */
int lzf_compress_batch_host(const lzf_compress_job* jobs, lzf_job_result* results, uint32_t n_jobs);
int lzf_decompress_batch_host(const lzf_decompress_job* jobs, lzf_job_result* results, uint32_t n_jobs);
/* EncoderTable::replace on a HOST table — src/raw/compress/mod.rs:19-25 (trait), :64-71 (U32Table), :88-96 (U16Table):
 * swaps `pos + table.offset` into the slot of hash(input[pos..]) and returns the previous entry minus table.offset,
 * saturating at 0 (stale entries read as position 0).  hash = hash_for_u32 (:40-51: 5 bytes of an 8-byte little-endian
 * read, slot 0 when fewer than 8 bytes remain) or hash_for_u16 (:58-61).  With lzf_table_offset's host twin below this
 * completes the trait for a crate that keeps `trait EncoderTable` and implements it on the C ABI's table structs.
 * Returns LZF_OK, or LZF_CONTRACT where the reference panics: pos + offset beyond the table's integer range (:67 / :92),
 * pos > input_len, or (U16) fewer than 4 bytes at pos (the slice read of :59). */
int lzf_table_replace_host(void* table, uint32_t table_kind, const uint8_t* input, uint64_t input_len, uint64_t pos,
                           uint64_t* previous);
/* EncoderTable::offset on a host table (mod.rs:72-74, :97-99). */
int lzf_table_offset_host(void* table, uint32_t table_kind, uint64_t add);

/* raw::compress2 with the reference's signature for ANY writer — src/raw/compress/mod.rs:165-166:
 *   pub fn compress2<W: Write, T: EncoderTable>(input: &[u8], cursor: usize, table: &mut T, mut writer: W)
 * `write_all(ctx, data, len)` stands for `writer.write_all(data)`: 0 = Ok(()), any other value = the writer's error, which
 * ends the call and is handed back in *writer_error (the Rust wrapper turns it into the io::Error again).  The block is
 * compressed on the device against LZ4's worst-case bound, then the reference's own sequence of write calls is replayed
 * from it — per sequence: the token byte, the literal length's tail (0xFF bytes four at a time, then one at a time, then
 * the remainder byte: mod.rs:243-260), the literals in one call, the two offset bytes, the match length's tail
 * (mod.rs:150-163); the final literal-only section likewise (:182-189).  A writer that refuses call k has seen exactly
 * the calls 0..k-1, as under the reference.  `table` (host memory, may be NULL = T::default() thrown away) ends in the
 * state the reference leaves it in: after the whole input, or — when the writer refuses — after the search of the
 * sequence whose write was refused (the reference mutates the table before it writes a sequence, :196-218 then :236).
 * Returns LZF_OK, LZF_OUTPUT_FULL (the writer refused), LZF_CONTRACT, or a negative library code.  Host buffers. */
typedef int (*lzf_write_all_fn)(void* ctx, const uint8_t* data, size_t len);
int lzf_compress2_host_writer(const uint8_t* input, uint64_t input_len, uint64_t cursor, void* table, uint32_t table_kind,
                              lzf_write_all_fn write_all, void* ctx, int* writer_error);

/* lzf_xxh32_batch over host buffers (staged to the device, hashed there; `out` is a host array). */
int lzf_xxh32_batch_host(const uint8_t* const* ptrs, const uint64_t* lens, uint32_t* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* LZFEAR_HIP_H */
