"""Decompress / compress rate on BASELINE config 5's data shape (256-byte motif repeated, 64 KiB blocks) — analysis."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
BS = 64 << 10
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
data = synth.repeat256(BS * 16)                       # 1 MiB pattern, tiled below
d_in = torch.from_numpy(np.tile(data, nblk // 16)).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
d_out = torch.empty(n * BS, dtype=torch.uint8, device='cuda'); d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
cj = device.to_device(blocks.compress_jobs(d_out, BS), 'cuda')
for it in range(2):
    torch.cuda.synchronize(); t = time.time(); device.compress_batch(cj, d_res, n); torch.cuda.synchronize(); dt = time.time() - t
print(f"compress  {n} x 64 KiB: {dt*1e3:.2f} ms  {n*BS/dt/2**30:.1f} GiB/s")
res = device.results_to_host(d_res, n).copy()
dj = np.zeros(n, dtype=device.DJOB); d_dec = torch.zeros(n * BS, dtype=torch.uint8, device='cuda')
dj['input'] = d_out.data_ptr() + np.arange(n, dtype=np.uint64) * BS; dj['input_len'] = res['out_len']
dj['out'] = d_dec.data_ptr() + np.arange(n, dtype=np.uint64) * BS; dj['out_cap'] = BS; dj['output_limit'] = BS
d_dj = device.to_device(dj, 'cuda'); d_res2 = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
for it in range(2):
    torch.cuda.synchronize(); t = time.time(); device.decompress_batch(d_dj, d_res2, n); torch.cuda.synchronize(); dt = time.time() - t
print(f"decompress {n} x 64 KiB (clen {int(res['out_len'][0])}): {dt*1e3:.2f} ms  {n*BS/dt/2**30:.1f} GiB/s; equal: {bool(torch.equal(d_dec, d_in))}")
