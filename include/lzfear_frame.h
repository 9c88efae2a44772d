/*
 * lzfear_frame.h — C ABI of the host-side LZ4 *frame* layer that drives the GPU block codec
 * (SURVEY.md §8 row f1).  It mirrors lz-fear's `framed` module:
 *
 *   CompressionSettings + compress / compress_with_size   src/framed/compress.rs:36-157
 *   compress_internal (header, block loop, EndMark)       src/framed/compress.rs:160-282
 *   LZ4FrameReader::new / decode_block / decompress_frame src/framed/decompress.rs:102-288
 *   Flags / BlockDescriptor                               src/framed/header.rs:8-81
 *   MAGIC / INCOMPRESSIBLE / WINDOW_SIZE                  src/framed/mod.rs:16-20
 *
 * The reference calls the block codec once per block (compress.rs:243, decompress.rs:248); here
 * all independent blocks of a frame go to the GPU in ONE batch (lzf_compress_batch /
 * lzf_decompress_batch of lzfear_hip.h); linked-block frames are inherently sequential and run
 * block after block with the table / window carried between calls.
 * Buffers are host memory.  No CPU codec: the calls fail with LZF_E_NO_DEVICE without a GPU.
 * Checksums: the header checksum (a few bytes) is hashed on the host; block checksums are computed on the device for
 * all blocks of a call in one launch; content checksums on the device for frames up to 32 MiB and on host worker
 * threads, overlapping the kernels, for longer ones (XXH32 is one serial chain per buffer).
 */
#ifndef LZFEAR_FRAME_H
#define LZFEAR_FRAME_H

#include <stddef.h>
#include <stdint.h>
#include "lzfear_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* frame-level status codes (>= 16; block-level DecodeError codes 1..4 of lzfear_hip.h pass through
 * as `CodecError`, src/framed/decompress.rs:20-21) */
enum {
    LZF_F_INPUT_ERROR = 16,            /* DecompressionError::InputError (EOF), decompress.rs:18-19 */
    LZF_F_WRONG_MAGIC = 17,            /* :24-25 */
    LZF_F_HEADER_CHECKSUM_FAIL = 18,   /* :26-27 */
    LZF_F_BLOCK_CHECKSUM_FAIL = 19,    /* :28-29 */
    LZF_F_FRAME_CHECKSUM_FAIL = 20,    /* :30-31 */
    LZF_F_BLOCK_LENGTH_OVERFLOW = 21,  /* :32-33 */
    LZF_F_BLOCK_SIZE_OVERFLOW = 22,    /* :34-35 */
    LZF_F_UNIMPLEMENTED_BLOCKSIZE = 23,/* header::ParseError, header.rs:19-28 */
    LZF_F_UNSUPPORTED_VERSION = 24,
    LZF_F_RESERVED_FLAG_BITS = 25,
    LZF_F_RESERVED_BD_BITS = 26,
    LZF_F_INVALID_BLOCK_SIZE = 27,     /* CompressionError::InvalidBlockSize, compress.rs:21-22 */
    LZF_F_PANIC = 28                   /* BlockDescriptor::new unwrap() panic, header.rs:55 */
};
/* per-frame status of the *_many calls only: the frame alone asks for more device memory than the budget allows */
#define LZF_E_NO_MEMORY (-4)

#define LZF_MAGIC 0x184D2204u          /* src/framed/mod.rs:16 */
#define LZF_WINDOW_SIZE 65536u         /* src/framed/mod.rs:20 */

/* CompressionSettings (src/framed/compress.rs:36-55) plus the `content_size: Option<u64>` of
 * compress_internal (:160; compress_with_size passes Some(len), :148-157). */
typedef struct lzf_settings {
    int32_t independent_blocks;     /* default 1  (:47) */
    int32_t block_checksums;        /* default 0  (:48) */
    int32_t content_checksum;       /* default 1  (:49) */
    int32_t has_dictionary_id;      /* dictionary(id, d) sets it; dictionary_id_nonsense_override clears/sets it (:113-133) */
    uint64_t block_size;            /* default 4 MiB (:50); 64 KiB / 256 KiB / 1 MiB / 4 MiB */
    const uint8_t* dictionary;      /* NULL = None (:51) */
    uint64_t dictionary_len;
    uint32_t dictionary_id;
    int32_t has_content_size;
    uint64_t content_size;
} lzf_settings;

void lzf_settings_default(lzf_settings* s);                       /* Default::default(), :44-55 */
size_t lzf_frame_compress_bound(const lzf_settings* s, size_t in_len);

/* CompressionSettings::compress / compress_with_size_unchecked over memory buffers.
 * Returns LZF_OK, LZF_F_INVALID_BLOCK_SIZE, LZF_F_PANIC, LZF_OUT_CAPACITY or a negative LZF_E_*. */
int lzf_frame_compress(const lzf_settings* s, const uint8_t* in, size_t in_len,
                       uint8_t* out, size_t out_cap, size_t* out_len);

/* Parsed frame header (LZ4FrameReader::new + accessors :167-175). */
typedef struct lzf_frame_info {
    uint8_t flags;                  /* FLG byte */
    uint8_t bd;                     /* BD byte */
    uint16_t header_len;            /* bytes up to and including HC */
    uint32_t dictionary_id;
    int32_t has_dictionary_id;
    int32_t has_content_size;
    uint64_t content_size;
    uint64_t block_maxsize;
} lzf_frame_info;
int lzf_frame_read_header(const uint8_t* in, size_t in_len, lzf_frame_info* info);

/* decompress_frame (:284-288) with an optional dictionary (into_read_with_dictionary, :180).
 * Status is the inner error kind (what decode_block returns).  *out_len = bytes produced by the
 * blocks completed before the error; *consumed = bytes of `in` read. */
int lzf_frame_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                         uint8_t* out, size_t out_cap, size_t* out_len, size_t* consumed);

/* Many frames per call: every block of every frame goes into the same launches and stays on the device from the
 * first block to the last (independent-block frames: one launch; linked-block frames: block k of every stream in
 * launch k, no host round trip in between).  Same bytes and the same per-frame statuses as one
 * lzf_frame_compress / lzf_frame_decompress call per frame; status[f], out_len[f] (and consumed[f], may be NULL) are
 * per frame, the return value is LZF_OK or a negative LZF_E_* (device trouble, bad arguments).  All frames of a
 * compress call share `s` (and its dictionary), all frames of a decompress call the dictionary. */
int lzf_frame_compress_many(const lzf_settings* s, uint32_t n_frames, const uint8_t* const* in, const size_t* in_len,
                            uint8_t* const* out, const size_t* out_cap, size_t* out_len, int* status);
int lzf_frame_decompress_many(uint32_t n_frames, const uint8_t* const* in, const size_t* in_len,
                              const uint8_t* dict, size_t dict_len,
                              uint8_t* const* out, const size_t* out_cap, size_t* out_len, size_t* consumed, int* status);

/* Block-by-block reader: LZ4FrameReader (src/framed/decompress.rs:79-282) over a memory buffer that the caller keeps
 * alive.  lzf_frame_reader_new = LZ4FrameReader::new (:102-161, header errors as its return value, *r = NULL then);
 * lzf_frame_reader_decode_block = decode_block (:198-282): decodes the next block into out[0..out_cap) — *out_len = 0
 * with LZF_OK once the EndMark has been read (`finished`, :206-215; the content checksum is verified there) —
 * with `dict` as the reference's `dictionary` argument (:238-245; linked frames carry their own 64 KiB window, :253-269).
 * One block per call means one small launch per call: the *_many drivers above are the fast path, this is the
 * reference's streaming interface for callers that want it.  Needs out_cap >= block_maxsize + the block's compressed
 * size to never see LZF_OUT_CAPACITY. */
typedef struct lzf_frame_reader lzf_frame_reader;
int lzf_frame_reader_new(const uint8_t* in, size_t in_len, lzf_frame_reader** r);
void lzf_frame_reader_free(lzf_frame_reader* r);
void lzf_frame_reader_info(const lzf_frame_reader* r, lzf_frame_info* info);     /* block_size(), frame_size(), dictionary_id() :167-175 */
int lzf_frame_reader_decode_block(lzf_frame_reader* r, const uint8_t* dict, size_t dict_len,
                                  uint8_t* out, size_t out_cap, size_t* out_len);
int lzf_frame_reader_finished(const lzf_frame_reader* r);
size_t lzf_frame_reader_consumed(const lzf_frame_reader* r);                     /* bytes of `in` read so far */

/* Streaming frame WRITER: CompressionSettings::compress / compress_with_size_unchecked (src/framed/compress.rs:138-157) for a
 * caller that feeds the stream piece by piece and takes the frame piece by piece — bounded memory, any stream length.
 * compress_internal's loop (:221-276) reads `block_size` bytes per turn; here the bytes come through
 * lzf_frame_writer_write in any granularity and leave through `write_all` (see lzf_compress2_host_writer), in the
 * reference's order: header (:163-200) at the first write or at finish, per block the length word, the payload and the
 * optional block checksum (:244-263), EndMark and content checksum (:277-281) at finish.
 * Independent blocks: `blocks_per_launch` whole blocks (0 = 64) are buffered and compressed in one launch (with a
 * dictionary: each block behind its own copy of it and the seeded template table, :217-220,:265-270).
 * Linked blocks: one launch per block, the table and the last 64 KiB carried (:271-275).
 * settings->has_content_size / content_size = compress_with_size_unchecked's argument.  The dictionary is copied.
 * Byte-identical to lzf_frame_compress over the concatenated input.
 * Returns LZF_OK, LZF_F_INVALID_BLOCK_SIZE / LZF_F_PANIC (new), LZF_OUTPUT_FULL when the sink refused a write (its code:
 * lzf_frame_writer_sink_error; the writer is dead from then on), LZF_CONTRACT, or a negative LZF_E_*. */
typedef struct lzf_frame_writer lzf_frame_writer;
int lzf_frame_writer_new(const lzf_settings* s, lzf_write_all_fn write_all, void* ctx, uint32_t blocks_per_launch, lzf_frame_writer** w);
int lzf_frame_writer_write(lzf_frame_writer* w, const uint8_t* data, size_t len);
int lzf_frame_writer_finish(lzf_frame_writer* w);
int lzf_frame_writer_sink_error(const lzf_frame_writer* w);
void lzf_frame_writer_free(lzf_frame_writer* w);

/* ---- the host side of the drivers (host_staging.h): one pinned slab, kept device scratch, worker threads ---------- */
typedef struct lzf_frame_stats {
    uint64_t calls;                     /* *_many calls served */
    uint64_t device_block_hashes;       /* block checksums computed by lzf_xxh32_batch */
    uint64_t host_block_hashes;         /* ... by the host (lzf_frame_assemble only: its payloads are host memory) */
    uint64_t device_content_hashes;     /* content checksums: one device chain per frame (frames <= 32 MiB) */
    uint64_t host_content_hashes;       /* ... on the worker threads (longer frames) */
    uint64_t h2d_copies, d2h_copies;    /* DMA transfers issued through the pinned slab, and their bytes */
    uint64_t h2d_bytes, d2h_bytes;
    uint64_t pinned_bytes;              /* size of the pinned slab now */
} lzf_frame_stats;
void lzf_frame_get_stats(lzf_frame_stats* st);
/* The drivers keep their pinned slab and device scratch between calls (allocating gigabytes costs more than the
 * kernels); this gives everything back.  Calls are serialised per device (one slab per device). */
void lzf_frame_release_scratch(void);
/* Worker threads of the host staging (pageable <-> pinned copies, content hashes of long frames): n of them, 0 = default (12 on a
 * large host), LZF_HOST_THREADS_NONE = none at all — the calling thread does every copy itself (no hidden host
 * threads are REQUIRED; they are a throughput option).  The threads are created on first use and kept. */
#define LZF_HOST_THREADS_NONE 0xFFFFFFFFu
void lzf_frame_set_host_threads(uint32_t n);
/* Device memory one pass of lzf_frame_decompress_many / lzf_frame_compress_many may use (0 = half of what is free): more frames
 * than fit are processed in several passes; a frame that does not fit alone gets status LZF_E_NO_MEMORY (decompress) or is a
 * pass of its own (compress). */
void lzf_frame_set_memory_budget(size_t bytes);
/* Pinned host memory the staging may hold (0 = default: 2 GiB; at least two 4 MiB slots).  A pass that moves more than this
 * recycles the slab as a ring of 4 MiB slots — a slot is reused when the DMA that last read it has finished — so the pinned
 * footprint of a call is bounded whatever the size of the call.  Changing it gives the current slab back. */
void lzf_frame_set_pinned_limit(size_t bytes);

/* XXH32 on the host (header / content checksums; twox-hash XxHash32 in the reference). */
uint32_t lzf_xxh32(const uint8_t* p, size_t len, uint32_t seed);
/* Streaming form (the content hasher of a block-by-block reader, src/framed/decompress.rs:89,276-278). */
typedef struct lzf_xxh32_state { uint32_t v[4]; uint8_t buf[16]; uint32_t fill; uint32_t seed; uint64_t total; } lzf_xxh32_state;
void lzf_xxh32_reset(lzf_xxh32_state* st, uint32_t seed);
void lzf_xxh32_update(lzf_xxh32_state* st, const uint8_t* p, size_t len);
uint32_t lzf_xxh32_digest(const lzf_xxh32_state* st);

/* Frame assembly from already-compressed blocks (what rank 0 does after the RCCL all-gather of a
 * block-sharded compression): writes header, then for every block
 * [u32 len | stored-bit][bytes][xxh32]?, then EndMark and content checksum.
 *   comp_len[i] == UINT32_MAX  => block i is stored raw (compress2 returned LZF_OUTPUT_FULL)
 *   payload[i] points at comp_len[i] compressed bytes, or at the raw block when stored.
 * `raw_len[i]` is the uncompressed length of block i. */
int lzf_frame_assemble(const lzf_settings* s, uint32_t n_blocks, const uint8_t* const* payload,
                       const uint32_t* comp_len, const uint32_t* raw_len, uint32_t content_xxh32,
                       uint8_t* out, size_t out_cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* LZFEAR_FRAME_H */
