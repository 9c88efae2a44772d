/* lzfear_dist.h — C ABI of the one exchange step of the block-sharded frame (BASELINE configs[3]).
 *
 * In independent-blocks mode (src/framed/compress.rs:47, :265-270) rank r of W compresses a contiguous range of a stream's
 * blocks with lzf_compress_batch and no communication.  Reassembling ONE frame — the `[u32 size word][payload]` sequence the
 * reference's block loop writes (src/framed/compress.rs:243-258), header in front (:163-200), EndMark behind (:277) — is one
 * exchange: the per-block sizes (ncclAllGather), then every rank's packed segment to every peer at its final offset in the
 * frame (grouped ncclSend / ncclRecv: the direct all-gather xGMI's point-to-point links offer, nothing padded, nothing through
 * the host).  This library (liblzfear_dist.so) is that step over RCCL; it links librccl and liblzfear_hip (lzf_copy_ranges).
 * The codec library itself stays free of the dependency.  Every rank calls every function collectively, on its own device.
 *
 * A Rust `src/framed` would bind these five functions in its -sys crate next to lzfear_hip.h / lzfear_frame.h. */
#ifndef LZFEAR_DIST_H
#define LZFEAR_DIST_H
#include <stddef.h>
#include <stdint.h>
#include "lzfear_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define LZF_DIST_UNIQUE_ID_BYTES 128
typedef struct lzf_dist_comm lzf_dist_comm;      /* an RCCL communicator + the rank / world it was made for */

/* ncclGetUniqueId: rank 0 calls this and ships the 128 bytes to the other ranks by whatever the host program has (the
 * launcher's store, MPI, a socket, torch.distributed's broadcast). */
int lzf_dist_unique_id(uint8_t id[LZF_DIST_UNIQUE_ID_BYTES]);
/* ncclCommInitRank on the calling thread's current HIP device.  world == 1 is allowed (the calls below then move nothing). */
int lzf_dist_comm_init(const uint8_t id[LZF_DIST_UNIQUE_ID_BYTES], int rank, int world, lzf_dist_comm** comm);
/* ncclCommCount: the ranks the communicator really spans (what bench.py reports as n_ranks_seen_by_rccl). */
int lzf_dist_comm_count(const lzf_dist_comm* comm);
void lzf_dist_comm_free(lzf_dist_comm* comm);

/* The exchange.  Rank r holds the results of its lzf_compress_batch over its n_local blocks — block i's input at
 * d_src + i * stride, its output slot at d_comp + i * stride, `stride` >= block_size — of a stream of n_blocks blocks of
 * block_size bytes (the stream's LAST block has last_block_len bytes); ranks hold contiguous ranges in rank order, the first
 * n_blocks % world ranks one block more (the partition of SURVEY §8(e)).  A block whose status is LZF_OUTPUT_FULL travels raw
 * with the stored bit in its size word (compress.rs:250-255); any other status fails the call on every rank.
 * On return (after the stream is synchronised by this call) d_frame[0 .. *frame_len) on EVERY rank is the frame:
 * header[0 .. header_len) (the caller's: lzf_frame_* builds it), the blocks in order, the EndMark (no content checksum: XXH32
 * does not compose across ranks).  *comp_total = the bytes of all payloads.
 * The header must not announce block checksums or a content checksum (this frame carries neither).
 * Returns LZF_OK, LZF_E_INVALID (arguments, frame_cap too small, such a header), LZF_E_HIP, or LZF_CONTRACT (a block status that
 * cannot be framed) — the SAME code on every rank, whichever rank the cause was on: a rank's local findings travel with the size
 * table, so no rank returns while its peers wait in the payload exchange. */
int lzf_frame_gather(lzf_dist_comm* comm, const lzf_job_result* d_results, const uint8_t* d_comp, const uint8_t* d_src,
                     uint64_t stride, uint64_t block_size, uint32_t n_local, uint32_t n_blocks, uint64_t last_block_len,
                     const uint8_t* header, uint32_t header_len, uint8_t* d_frame, uint64_t frame_cap,
                     uint64_t* frame_len, uint64_t* comp_total, void* hip_stream);

const char* lzf_dist_last_error(void);
/* Path of the librccl this library's calls resolve to in the running process (dladdr): a host program that carries its own RCCL
 * (PyTorch does) can check that both bind the same one. */
const char* lzf_dist_rccl_path(void);

#ifdef __cplusplus
}
#endif
#endif /* LZFEAR_DIST_H */
