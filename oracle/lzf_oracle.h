/*
 * lzf_oracle.h — CPU restatement of lz-fear's LZ4 raw-block codec and frame layer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rust-lz-fear_amd/, include/)
 * may include, link, import or execute anything under oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker
 * (or as the timed CPU baseline), never as the thing shipped.
 *
 * What it restates (all citations are file:line into the lz-fear 0.2.0 sources):
 *   src/raw/compress/mod.rs      :19-101  EncoderTable / U32Table / U16Table / hashes
 *                                :117-145 count_matching_bytes
 *                                :150-163 write_group,  :239-260 LSIC coding
 *                                :165-238 compress2
 *   src/raw/decompress.rs        :7-26    DecodeError,  :30-43 read_lsic
 *                                :58-78   decompress_raw, :80-138 copy_overlapping
 *   src/framed/compress.rs       :160-282 compress_internal, :294-314 NoPartialWrites
 *   src/framed/decompress.rs     :102-161 LZ4FrameReader::new, :198-279 decode_block
 *   src/framed/header.rs         :30-81   Flags / BlockDescriptor
 *   src/framed/mod.rs            :16-20   MAGIC / INCOMPRESSIBLE / WINDOW_SIZE
 *   XXH32: twox-hash 1.6.x (Cargo.toml:17; not vendored) — restated from the public XXH32
 *   specification; pinned by the reference's own fuzz-corpus frames (self-checking content
 *   checksums) and cross-checked against the python `xxhash` module in tests.
 *
 * Parity pinning (see DESIGN.md "Oracle"): the Rust reference cannot be built here (no
 * cargo/rustc, dependencies not vendored).  The oracle is pinned by
 *   (1) the reference's decode KATs (src/raw/decompress.rs:153-175),
 *   (2) the reference's fuzz corpus (fuzz/corpus/decode: 3 valid self-checking frames + the
 *       error-class census of SURVEY.md §4) — read in the build container only,
 *   (3) tests/issue-15.rs data (linked 64 KiB round trip),
 *   (4) liblz4 1.9.3 in the regimes where lz-fear's README (README.md:5,14-16) and
 *       tests/output_equivalence.rs claim byte equality (U16 table == LZ4_compress_default for
 *       inputs < 64 KiB; U32 table == LZ4_compress_fast_continue on a fresh stream),
 *   (5) the formula-defined KAT-A / KAT-B and fingerprints G1..G5 of SURVEY.md App. B/C.
 */
#ifndef LZF_ORACLE_H
#define LZF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (shared numbering with include/lzfear_hip.h) ---- */
enum {
    LZFO_OK = 0,
    /* raw::DecodeError, src/raw/decompress.rs:8-17 */
    LZFO_UNEXPECTED_END = 1,
    LZFO_MEMORY_LIMIT_EXCEEDED = 2,
    LZFO_ZERO_DEDUP_OFFSET = 3,
    LZFO_INVALID_DEDUP_OFFSET = 4,
    /* compress2: the writer refused a write (NoPartialWrites -> ConnectionAborted) */
    LZFO_OUTPUT_FULL = 5,
    /* contract violations that panic in the reference (mod.rs:167, :67/:92) */
    LZFO_CONTRACT = 6,
    /* caller's out buffer too small to hold what the reference would have produced */
    LZFO_OUT_CAPACITY = 7,
    /* framed::DecompressionError, src/framed/decompress.rs:17-36 */
    LZFO_F_INPUT_ERROR = 16,          /* io::Error (always EOF here) */
    LZFO_F_WRONG_MAGIC = 17,
    LZFO_F_HEADER_CHECKSUM_FAIL = 18,
    LZFO_F_BLOCK_CHECKSUM_FAIL = 19,
    LZFO_F_FRAME_CHECKSUM_FAIL = 20,
    LZFO_F_BLOCK_LENGTH_OVERFLOW = 21,
    LZFO_F_BLOCK_SIZE_OVERFLOW = 22,
    /* header::ParseError, src/framed/header.rs:19-28 */
    LZFO_F_UNIMPLEMENTED_BLOCKSIZE = 23,
    LZFO_F_UNSUPPORTED_VERSION = 24,
    LZFO_F_RESERVED_FLAG_BITS = 25,
    LZFO_F_RESERVED_BD_BITS = 26,
    /* framed::CompressionError::InvalidBlockSize, src/framed/compress.rs:22 */
    LZFO_F_INVALID_BLOCK_SIZE = 27,
    /* BlockDescriptor::new panics (header.rs:55 unwrap) for 0 and >= 16 MiB powers */
    LZFO_F_PANIC = 28
};

#define LZFO_TABLE_U32 0
#define LZFO_TABLE_U16 1

typedef struct { uint32_t dict[4096]; uint64_t offset; } lzfo_u32_table;   /* mod.rs:28-31 */
typedef struct { uint16_t dict[8192]; uint64_t offset; } lzfo_u16_table;   /* mod.rs:79-82 */

/* EncoderTable::replace / ::offset (mod.rs:63-76, :88-101).  replace returns the previous
 * position; *contract is set to 1 when the reference would panic. */
size_t lzfo_u32_replace(lzfo_u32_table* t, const uint8_t* input, size_t len, size_t pos, int* contract);
size_t lzfo_u16_replace(lzfo_u16_table* t, const uint8_t* input, size_t len, size_t pos, int* contract);

/* compress2 (mod.rs:165-238).  `table` is lzfo_u32_table* or lzfo_u16_table* per `kind`.
 * The sink has NoPartialWrites semantics with capacity `cap` (framed/compress.rs:294-314):
 * a write that does not fit fails and compress2 returns LZFO_OUTPUT_FULL. */
int lzfo_compress2(const uint8_t* input, size_t len, size_t cursor, int kind, void* table,
                   uint8_t* out, size_t cap, size_t* out_len);

/* decompress_raw (decompress.rs:58-78).  `out` holds `*out_len` bytes of addressable history
 * on entry (the Vec's existing content) and `out_cap` bytes of room in total. */
int lzfo_decompress_raw(const uint8_t* input, size_t len, const uint8_t* prefix, size_t prefix_len,
                        uint8_t* out, size_t* out_len, size_t out_cap, size_t output_limit);

uint32_t lzfo_xxh32(const uint8_t* p, size_t len, uint32_t seed);

/* CompressionSettings (framed/compress.rs:36-55) + the content_size Option of
 * compress_internal (:160). */
typedef struct {
    int independent_blocks;      /* default 1 */
    int block_checksums;         /* default 0 */
    int content_checksum;        /* default 1 */
    uint64_t block_size;         /* default 4 MiB */
    const uint8_t* dictionary;   /* NULL = None */
    uint64_t dictionary_len;
    int has_dictionary_id;
    uint32_t dictionary_id;
    int has_content_size;
    uint64_t content_size;
} lzfo_settings;

void lzfo_settings_default(lzfo_settings* s);

/* CompressionSettings::compress* over an in-memory reader/writer. */
int lzfo_frame_compress(const lzfo_settings* s, const uint8_t* in, size_t in_len,
                        uint8_t* out, size_t out_cap, size_t* out_len);
size_t lzfo_frame_compress_bound(const lzfo_settings* s, size_t in_len);

/* decompress_frame / LZ4FrameReader + read_to_end with a dictionary
 * (framed/decompress.rs:102-161,198-279,284-288).  On error *out_len holds the bytes the
 * reader had already handed out (completed blocks). *consumed = bytes of `in` read. */
int lzfo_frame_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                          uint8_t* out, size_t out_cap, size_t* out_len, size_t* consumed);

#ifdef __cplusplus
}
#endif
#endif
