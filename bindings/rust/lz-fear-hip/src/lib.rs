//! Safe wrappers keeping the reference's signatures.  UNVERIFIED SOURCE (no Rust toolchain here).
pub mod raw {
    use lz_fear_hip_sys as sys;
    use std::io::{self, ErrorKind, Write};
    pub use lz_fear::raw::DecodeError; // src/raw/decompress.rs:7-17

    /// The two table types the kernels know (`trait EncoderTable`, src/raw/compress/mod.rs:19-25).
    pub trait GpuTable: Default + Clone {
        const KIND: u32;
        fn payload_size_limit() -> usize;
        fn as_mut_ptr(&mut self) -> *mut std::ffi::c_void;
        fn offset(&mut self, n: usize);
    }
    #[repr(transparent)]
    pub struct U32Table(Box<sys::lzf_u32_table>);
    #[repr(transparent)]
    pub struct U16Table(Box<sys::lzf_u16_table>);
    impl Default for U32Table { fn default() -> Self { U32Table(Box::new(sys::lzf_u32_table { dict: [0; 4096], offset: 0 })) } }
    impl Default for U16Table { fn default() -> Self { U16Table(Box::new(sys::lzf_u16_table { dict: [0; 8192], offset: 0 })) } }
    impl Clone for U32Table { fn clone(&self) -> Self { U32Table(Box::new(sys::lzf_u32_table { dict: self.0.dict, offset: self.0.offset })) } }
    impl Clone for U16Table { fn clone(&self) -> Self { U16Table(Box::new(sys::lzf_u16_table { dict: self.0.dict, offset: self.0.offset })) } }
    impl GpuTable for U32Table {
        const KIND: u32 = sys::LZF_TABLE_U32;
        fn payload_size_limit() -> usize { u32::MAX as usize } // mod.rs:75
        fn as_mut_ptr(&mut self) -> *mut std::ffi::c_void { &mut *self.0 as *mut _ as *mut _ }
        fn offset(&mut self, n: usize) { self.0.offset += n as u64 } // mod.rs:72-74
    }
    impl GpuTable for U16Table {
        const KIND: u32 = sys::LZF_TABLE_U16;
        fn payload_size_limit() -> usize { u16::MAX as usize } // mod.rs:100
        fn as_mut_ptr(&mut self) -> *mut std::ffi::c_void { &mut *self.0 as *mut _ as *mut _ }
        fn offset(&mut self, n: usize) { self.0.offset += n as u64 } // mod.rs:97-99
    }

    /// `raw::compress2` (src/raw/compress/mod.rs:165-166).  `cap` = what the writer can still take
    /// (the frame layer passes the payload length, src/framed/compress.rs:242; `usize::MAX` for a Vec).
    pub fn compress2<W: Write, T: GpuTable>(input: &[u8], cursor: usize, table: &mut T, mut writer: W, cap: usize) -> io::Result<()> {
        assert!(input.len() <= T::payload_size_limit()); // mod.rs:167
        let mut out = vec![0u8; cap.min(input.len() + input.len() / 255 + 16)];
        let job = sys::lzf_compress_job {
            input: input.as_ptr(), input_len: input.len() as u64, cursor: cursor as u64,
            out: out.as_mut_ptr(), out_cap: out.len() as u64,
            table: table.as_mut_ptr(), table_kind: T::KIND, flags: 0,
        };
        let mut res = sys::lzf_job_result::default();
        let rc = unsafe { sys::lzf_compress_batch_host(&job, &mut res, 1) };
        if rc != 0 { return Err(io::Error::new(ErrorKind::Other, "lzfear_hip: no device / HIP error")); }
        match res.status {
            sys::LZF_OK => writer.write_all(&out[..res.out_len as usize]),
            sys::LZF_OUTPUT_FULL => Err(ErrorKind::ConnectionAborted.into()), // NoPartialWrites, framed/compress.rs:300
            _ => panic!("EncoderTable contract violated"), // mod.rs:67
        }
    }

    /// `raw::decompress_raw` (src/raw/decompress.rs:58-59): appends to `output`, which is also history.
    pub fn decompress_raw(input: &[u8], prefix: &[u8], output: &mut Vec<u8>, output_limit: usize) -> Result<(), DecodeError> {
        let existing = output.len();
        // room for what this call can append: at most 255 bytes per input byte (a run-length byte is the densest code,
        // raw/decompress.rs:40-56) and at most up to the limit plus the literals that may overshoot it —
        // never `output_limit` itself, which callers set to usize::MAX-like values
        let grow = input.len().saturating_mul(255).saturating_add(16)
            .min(output_limit.saturating_sub(existing).saturating_add(input.len()));
        output.resize(existing + grow, 0);
        let job = sys::lzf_decompress_job {
            input: input.as_ptr(), input_len: input.len() as u64,
            prefix: prefix.as_ptr(), prefix_len: prefix.len() as u64,
            out: output.as_mut_ptr(), out_existing_len: existing as u64, out_cap: output.len() as u64,
            output_limit: output_limit as u64,
        };
        let mut res = sys::lzf_job_result::default();
        let rc = unsafe { sys::lzf_decompress_batch_host(&job, &mut res, 1) };
        assert_eq!(rc, 0, "lzfear_hip: no device / HIP error");
        output.truncate((res.out_len as usize).min(output.len()).max(existing));
        match res.status {
            sys::LZF_OK => Ok(()),
            sys::LZF_UNEXPECTED_END => Err(DecodeError::UnexpectedEnd),
            sys::LZF_MEMORY_LIMIT_EXCEEDED => Err(DecodeError::MemoryLimitExceeded),
            sys::LZF_ZERO_DEDUP_OFFSET => Err(DecodeError::ZeroDeduplicationOffset),
            sys::LZF_INVALID_DEDUP_OFFSET => Err(DecodeError::InvalidDeduplicationOffset),
            _ => panic!("lzfear_hip: contract / capacity"),
        }
    }
}
