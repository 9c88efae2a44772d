"""Device-resident batches for the C ABI: job arrays built with numpy, kept in HBM as torch
uint8 tensors.  torch is plumbing here (device memory + streams), not the product: the codec
is liblzfear_hip.so and every launch goes through its extern "C" entry points."""
import numpy as np
import torch

from . import ffi

CJOB = np.dtype([("input", "<u8"), ("input_len", "<u8"), ("cursor", "<u8"), ("out", "<u8"),
                 ("out_cap", "<u8"), ("table", "<u8"), ("table_kind", "<u4"), ("flags", "<u4")])
DJOB = np.dtype([("input", "<u8"), ("input_len", "<u8"), ("prefix", "<u8"), ("prefix_len", "<u8"),
                 ("out", "<u8"), ("out_existing_len", "<u8"), ("out_cap", "<u8"), ("output_limit", "<u8")])
RES = np.dtype([("out_len", "<u8"), ("status", "<i4"), ("reserved", "<u4")])
assert CJOB.itemsize == 56 and DJOB.itemsize == 64 and RES.itemsize == 16


def to_device(arr, device):
    """numpy (structured) array -> uint8 tensor in HBM."""
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    return torch.from_numpy(raw.copy()).to(device)


def results_to_host(d_res, n):
    return d_res.cpu().numpy().view(RES)[:n]


def _stream_ptr(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return stream.cuda_stream


def compress_batch(d_jobs, d_res, n, kinds=ffi.KINDS_U32, stream=None):
    ffi.check(ffi.lib().lzf_compress_batch(d_jobs.data_ptr(), d_res.data_ptr(), n, kinds, _stream_ptr(stream)))


def decompress_batch(d_jobs, d_res, n, stream=None, max_input_len=None):
    """lzf_decompress_batch; with max_input_len (an upper bound of the jobs' input_len, which the caller of a device-resident
    job array usually knows) lzf_decompress_batch_sized: the segmented pipeline then sizes its scratch for that bound instead
    of for 4 MiB blocks at their worst case — a batch of small blocks allocates (and launches) next to nothing."""
    if max_input_len is None:
        ffi.check(ffi.lib().lzf_decompress_batch(d_jobs.data_ptr(), d_res.data_ptr(), n, _stream_ptr(stream)))
    else:
        ffi.check(ffi.lib().lzf_decompress_batch_sized(d_jobs.data_ptr(), d_res.data_ptr(), n, int(max_input_len), _stream_ptr(stream)))


def xxh32_batch(d_ptrs, d_lens, d_out, n, stream=None):
    ffi.check(ffi.lib().lzf_xxh32_batch(d_ptrs.data_ptr(), d_lens.data_ptr(), d_out.data_ptr(), n, _stream_ptr(stream)))


def copy_ranges(d_src_ptrs, d_dst_ptrs, d_lens, n, max_len, stream=None):
    """Stored blocks moved on the device (lzf_copy_ranges); the three arrays are uint64 device tensors."""
    ffi.check(ffi.lib().lzf_copy_ranges(d_src_ptrs.data_ptr(), d_dst_ptrs.data_ptr(), d_lens.data_ptr(), n, max_len, _stream_ptr(stream)))


class BlockSet:
    """Equal-size independent blocks of one contiguous HBM buffer (the DP unit of
    src/framed/compress.rs:221-276 in independent-blocks mode): block i = data[i*bs : ...]."""

    def __init__(self, data, block_size):
        assert data.dtype == torch.uint8 and data.is_cuda and data.dim() == 1
        self.data = data
        self.block_size = block_size
        self.total = data.numel()
        self.n = (self.total + block_size - 1) // block_size
        self.lens = np.full(self.n, block_size, dtype=np.uint64)
        if self.total % block_size:
            self.lens[-1] = self.total % block_size
        self.offsets = np.arange(self.n, dtype=np.uint64) * np.uint64(block_size)

    def compress_jobs(self, out, out_stride):
        """One compress2 job per block, cursor 0, fresh U32Table, writer cap = block length
        (src/framed/compress.rs:242-243).  `out` = HBM slab of n * out_stride bytes."""
        j = np.zeros(self.n, dtype=CJOB)
        j["input"] = np.uint64(self.data.data_ptr()) + self.offsets
        j["input_len"] = self.lens
        j["out"] = np.uint64(out.data_ptr()) + np.arange(self.n, dtype=np.uint64) * np.uint64(out_stride)
        j["out_cap"] = self.lens
        j["table_kind"] = ffi.TABLE_U32
        return j
