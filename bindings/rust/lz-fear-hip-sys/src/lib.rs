//! FFI declarations of `include/lzfear_hip.h` (ABI version 1).  UNVERIFIED SOURCE: no Rust toolchain
//! was available where this was written; field order and sizes mirror the C header
//! (`lzf_compress_job` 56 B, `lzf_decompress_job` 64 B, `lzf_job_result` 16 B on LP64).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const LZF_OK: i32 = 0;
pub const LZF_UNEXPECTED_END: i32 = 1; // DecodeError::UnexpectedEnd            src/raw/decompress.rs:9-10
pub const LZF_MEMORY_LIMIT_EXCEEDED: i32 = 2; // DecodeError::MemoryLimitExceeded      :11-12
pub const LZF_ZERO_DEDUP_OFFSET: i32 = 3; // DecodeError::ZeroDeduplicationOffset  :13-14
pub const LZF_INVALID_DEDUP_OFFSET: i32 = 4; // DecodeError::InvalidDeduplicationOffset :15-16
pub const LZF_OUTPUT_FULL: i32 = 5; // writer error (NoPartialWrites -> ConnectionAborted)
pub const LZF_CONTRACT: i32 = 6; // the reference would panic (mod.rs:167, :67, :92)
pub const LZF_OUT_CAPACITY: i32 = 7;
pub const LZF_E_NO_DEVICE: c_int = -1;
pub const LZF_E_HIP: c_int = -2;
pub const LZF_E_INVALID: c_int = -3;
pub const LZF_TABLE_U32: u32 = 0;
pub const LZF_TABLE_U16: u32 = 1;
pub const LZF_KINDS_U32: u32 = 1;
pub const LZF_KINDS_U16: u32 = 2;
pub const LZF_CJOB_TABLE_READONLY: u32 = 1;

#[repr(C)]
pub struct lzf_u32_table { pub dict: [u32; 4096], pub offset: u64 } // src/raw/compress/mod.rs:27-31
#[repr(C)]
pub struct lzf_u16_table { pub dict: [u16; 8192], pub offset: u64 } // :78-82

#[repr(C)]
pub struct lzf_compress_job {
    pub input: *const u8,
    pub input_len: u64,
    pub cursor: u64,
    pub out: *mut u8,
    pub out_cap: u64,
    pub table: *mut c_void,
    pub table_kind: u32,
    pub flags: u32,
}

#[repr(C)]
pub struct lzf_decompress_job {
    pub input: *const u8,
    pub input_len: u64,
    pub prefix: *const u8,
    pub prefix_len: u64,
    pub out: *mut u8,
    pub out_existing_len: u64,
    pub out_cap: u64,
    pub output_limit: u64,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct lzf_job_result { pub out_len: u64, pub status: i32, pub reserved: u32 }

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct lzf_chain_state { pub length: u64, pub dead: u32, pub reserved: u32 }
#[repr(C)]
pub struct lzf_chain_step {
    pub prev_job: u32,
    pub job: u32,
    pub stored_len: u64,
    pub stored_src: *const u8,
    pub out: *mut u8,
    pub block_maxsize: u64,
}

/// CompressionSettings + content size (lzfear_frame.h; src/framed/compress.rs:36-55)
#[repr(C)] pub struct lzf_settings {
    pub independent_blocks: i32, pub block_checksums: i32, pub content_checksum: i32, pub has_dictionary_id: i32,
    pub block_size: u64, pub dictionary: *const u8, pub dictionary_len: u64, pub dictionary_id: u32, pub has_content_size: i32,
    pub content_size: u64,
}
/// `writer.write_all(data)`: 0 = Ok(()), anything else = the writer's error
pub type lzf_write_all_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, data: *const u8, len: usize) -> c_int>;
#[repr(C)] pub struct lzf_frame_writer { _private: [u8; 0] }
extern "C" {
    pub fn lzf_abi_version() -> c_int;
    pub fn lzf_last_error() -> *const c_char;
    pub fn lzf_device_count() -> c_int;
    pub fn lzf_compress_batch(d_jobs: *const lzf_compress_job, d_results: *mut lzf_job_result, n_jobs: u32,
                              table_kinds: u32, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_decompress_batch(d_jobs: *const lzf_decompress_job, d_results: *mut lzf_job_result, n_jobs: u32,
                                hip_stream: *mut c_void) -> c_int;
    pub fn lzf_compress_batch_host(jobs: *const lzf_compress_job, results: *mut lzf_job_result, n_jobs: u32) -> c_int;
    pub fn lzf_decompress_batch_host(jobs: *const lzf_decompress_job, results: *mut lzf_job_result, n_jobs: u32) -> c_int;
    pub fn lzf_table_seed_from_dictionary(d_table: *mut lzf_u32_table, d_dict: *const u8, dict_len: u64,
                                          hip_stream: *mut c_void) -> c_int;
    pub fn lzf_table_offset(d_table: *mut c_void, table_kind: u32, add: u64, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_table_offset_batch(d_tables: *const *mut c_void, d_adds: *const u64, n: u32, table_kind: u32, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_chain_decompress_step(d_steps: *const lzf_chain_step, d_state: *mut lzf_chain_state, n_streams: u32,
                                     d_jobs: *mut lzf_decompress_job, d_results: *const lzf_job_result, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_xxh32_batch(d_ptrs: *const *const u8, d_lens: *const u64, d_out: *mut u32, n: u32,
                           hip_stream: *mut c_void) -> c_int;
    pub fn lzf_copy_ranges(d_src: *const *const u8, d_dst: *const *mut u8, d_len: *const u64, n: u32, max_len: u64, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_xxh32_batch_host(ptrs: *const *const u8, lens: *const u64, out: *mut u32, n: u32) -> c_int;
    pub fn lzf_decompress_batch_sized(d_jobs: *const lzf_decompress_job, d_results: *mut lzf_job_result, n_jobs: u32, max_input_len: u64, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_last_decompress_launch() -> *const c_char;
    pub fn lzf_last_compress_launch() -> *const c_char;
    // EncoderTable::replace / ::offset on host tables (src/raw/compress/mod.rs:64-74, :88-99)
    pub fn lzf_table_replace_host(table: *mut c_void, table_kind: u32, input: *const u8, input_len: u64, pos: u64, previous: *mut u64) -> c_int;
    pub fn lzf_table_offset_host(table: *mut c_void, table_kind: u32, add: u64) -> c_int;
    // compress2 for any writer (mod.rs:165-166): the reference's write calls replayed into `write_all`
    pub fn lzf_compress2_host_writer(input: *const u8, input_len: u64, cursor: u64, table: *mut c_void, table_kind: u32,
                                     write_all: lzf_write_all_fn, ctx: *mut c_void, writer_error: *mut c_int) -> c_int;
    // streaming frame writer (src/framed/compress.rs:138-157, :221-276)
    pub fn lzf_frame_writer_new(s: *const lzf_settings, write_all: lzf_write_all_fn, ctx: *mut c_void, blocks_per_launch: u32, w: *mut *mut lzf_frame_writer) -> c_int;
    pub fn lzf_frame_writer_write(w: *mut lzf_frame_writer, data: *const u8, len: usize) -> c_int;
    pub fn lzf_frame_writer_finish(w: *mut lzf_frame_writer) -> c_int;
    pub fn lzf_frame_writer_sink_error(w: *const lzf_frame_writer) -> c_int;
    pub fn lzf_frame_writer_free(w: *mut lzf_frame_writer);
    // lzfear_frame.h: the block-by-block reader (LZ4FrameReader::new + decode_block) and the staging controls
    pub fn lzf_frame_reader_new(input: *const u8, in_len: usize, r: *mut *mut lzf_frame_reader) -> c_int;
    pub fn lzf_frame_reader_free(r: *mut lzf_frame_reader);
    pub fn lzf_frame_reader_decode_block(r: *mut lzf_frame_reader, dict: *const u8, dict_len: usize,
                                         out: *mut u8, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn lzf_frame_reader_finished(r: *const lzf_frame_reader) -> c_int;
    pub fn lzf_frame_reader_consumed(r: *const lzf_frame_reader) -> usize;
    pub fn lzf_frame_release_scratch();
    pub fn lzf_frame_set_host_threads(n: u32);
    pub fn lzf_frame_set_memory_budget(bytes: usize);
    pub fn lzf_frame_set_pinned_limit(bytes: usize);
}

// lzfear_dist.h (liblzfear_dist.so; links librccl): the one exchange of the block-sharded frame — what a multi-GPU `src/framed` calls behind
// its per-rank lzf_compress_batch (src/framed/compress.rs:243-258 for blocks compressed on different GPUs).  UNVERIFIED source, like the rest of
// this crate: there is no Rust toolchain in the build image.
#[repr(C)] pub struct lzf_dist_comm { _private: [u8; 0] }
pub const LZF_DIST_UNIQUE_ID_BYTES: usize = 128;
#[link(name = "lzfear_dist")]
extern "C" {
    pub fn lzf_dist_unique_id(id: *mut u8) -> c_int;
    pub fn lzf_dist_comm_init(id: *const u8, rank: c_int, world: c_int, comm: *mut *mut lzf_dist_comm) -> c_int;
    pub fn lzf_dist_comm_count(comm: *const lzf_dist_comm) -> c_int;
    pub fn lzf_dist_comm_free(comm: *mut lzf_dist_comm);
    pub fn lzf_frame_gather(comm: *mut lzf_dist_comm, d_results: *const lzf_job_result, d_comp: *const u8, d_src: *const u8,
                            stride: u64, block_size: u64, n_local: u32, n_blocks: u32, last_block_len: u64,
                            header: *const u8, header_len: u32, d_frame: *mut u8, frame_cap: u64,
                            frame_len: *mut u64, comp_total: *mut u64, hip_stream: *mut c_void) -> c_int;
    pub fn lzf_dist_last_error() -> *const c_char;
}
