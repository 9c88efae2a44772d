//! Safe wrappers keeping the reference's signatures.  UNVERIFIED SOURCE (no Rust toolchain here).
pub mod raw {
    use lz_fear_hip_sys as sys;
    use std::io::{self, ErrorKind, Write};
    pub use lz_fear::raw::DecodeError; // src/raw/decompress.rs:7-17

    /// `trait EncoderTable` (src/raw/compress/mod.rs:19-25) — all three methods — for the two table types the kernels know.
    pub trait EncoderTable {
        fn payload_size_limit() -> usize;
        fn replace(&mut self, input: &[u8], offset: usize) -> usize;
        fn offset(&mut self, offset: usize);
    }
    /// What the C ABI needs on top of the trait: the table's kind and its address.
    pub trait GpuTable: EncoderTable + Default + Clone {
        const KIND: u32;
        fn as_mut_ptr(&mut self) -> *mut std::ffi::c_void;
    }
    fn replace_via_abi(table: *mut std::ffi::c_void, kind: u32, input: &[u8], offset: usize) -> usize {
        let mut prev = 0u64;
        let rc = unsafe { sys::lzf_table_replace_host(table, kind, input.as_ptr(), input.len() as u64, offset as u64, &mut prev) };
        assert_eq!(rc, sys::LZF_OK, "EncoderTable contract violated"); // mod.rs:67 / :92
        prev as usize
    }
    #[repr(transparent)]
    pub struct U32Table(Box<sys::lzf_u32_table>);
    #[repr(transparent)]
    pub struct U16Table(Box<sys::lzf_u16_table>);
    impl Default for U32Table { fn default() -> Self { U32Table(Box::new(sys::lzf_u32_table { dict: [0; 4096], offset: 0 })) } }
    impl Default for U16Table { fn default() -> Self { U16Table(Box::new(sys::lzf_u16_table { dict: [0; 8192], offset: 0 })) } }
    impl Clone for U32Table { fn clone(&self) -> Self { U32Table(Box::new(sys::lzf_u32_table { dict: self.0.dict, offset: self.0.offset })) } }
    impl Clone for U16Table { fn clone(&self) -> Self { U16Table(Box::new(sys::lzf_u16_table { dict: self.0.dict, offset: self.0.offset })) } }
    impl EncoderTable for U32Table {
        fn payload_size_limit() -> usize { u32::MAX as usize } // mod.rs:75
        fn replace(&mut self, input: &[u8], offset: usize) -> usize { replace_via_abi(self.as_mut_ptr(), sys::LZF_TABLE_U32, input, offset) } // mod.rs:64-71
        fn offset(&mut self, n: usize) { self.0.offset += n as u64 } // mod.rs:72-74
    }
    impl EncoderTable for U16Table {
        fn payload_size_limit() -> usize { u16::MAX as usize } // mod.rs:100
        fn replace(&mut self, input: &[u8], offset: usize) -> usize { replace_via_abi(self.as_mut_ptr(), sys::LZF_TABLE_U16, input, offset) } // mod.rs:88-96
        fn offset(&mut self, n: usize) { self.0.offset += n as u64 } // mod.rs:97-99
    }
    impl GpuTable for U32Table {
        const KIND: u32 = sys::LZF_TABLE_U32;
        fn as_mut_ptr(&mut self) -> *mut std::ffi::c_void { &mut *self.0 as *mut _ as *mut _ }
    }
    impl GpuTable for U16Table {
        const KIND: u32 = sys::LZF_TABLE_U16;
        fn as_mut_ptr(&mut self) -> *mut std::ffi::c_void { &mut *self.0 as *mut _ as *mut _ }
    }

    /// `raw::compress2` with the reference's signature (src/raw/compress/mod.rs:165-166): any `W: Write`, no capacity
    /// argument.  The block is compressed on the device against LZ4's worst-case bound; lzf_compress2_host_writer then
    /// replays the reference's own write calls (token, length tail, literals, offset, length tail: mod.rs:150-163,
    /// :243-260) into `writer` through the trampoline below and stops at the first call the writer refuses — the error
    /// returned is the writer's own, and `table` is left as the reference leaves it at that point.
    ///
    /// One job per call = one block at single-block latency (hundreds of ms for 4 MiB): a port of `src/framed` must call
    /// the `_many` / batch entry points (lzf_frame_compress_many, lzf_frame_writer_*), never this wrapper in a loop.
    pub fn compress2<W: Write, T: GpuTable>(input: &[u8], cursor: usize, table: &mut T, mut writer: W) -> io::Result<()> {
        assert!(input.len() <= T::payload_size_limit()); // mod.rs:167
        struct Sink<'a> { w: &'a mut dyn Write, err: Option<io::Error>, panic: Option<Box<dyn std::any::Any + Send>> }
        // A panic of the user's writer must not unwind through the C frames of lzf_compress2_host_writer (undefined
        // behaviour): it is caught here, the writer "refuses" (non-zero), and the panic resumes once the FFI call has
        // returned — the caller sees what the reference's compress2 would show it: the writer's own panic.
        unsafe extern "C" fn write_all(ctx: *mut std::ffi::c_void, data: *const u8, len: usize) -> i32 {
            let s = &mut *(ctx as *mut Sink);
            let bytes = std::slice::from_raw_parts(data, len);
            match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| s.w.write_all(bytes))) {
                Ok(Ok(())) => 0,
                Ok(Err(e)) => { s.err = Some(e); 1 }
                Err(p) => { s.panic = Some(p); 2 }
            }
        }
        let mut sink = Sink { w: &mut writer, err: None, panic: None };
        let mut werr = 0i32;
        let rc = unsafe {
            sys::lzf_compress2_host_writer(input.as_ptr(), input.len() as u64, cursor as u64, table.as_mut_ptr(), T::KIND,
                                           Some(write_all), &mut sink as *mut _ as *mut _, &mut werr)
        };
        if let Some(p) = sink.panic.take() { std::panic::resume_unwind(p) }
        match rc {
            sys::LZF_OK => Ok(()),
            sys::LZF_OUTPUT_FULL => Err(sink.err.take().unwrap_or_else(|| ErrorKind::ConnectionAborted.into())), // e.g. NoPartialWrites, framed/compress.rs:300
            sys::LZF_CONTRACT => panic!("EncoderTable contract violated"), // mod.rs:67
            _ => Err(io::Error::new(ErrorKind::Other, "lzfear_hip: no device / HIP error")),
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
