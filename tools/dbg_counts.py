"""Per-block parse statistics of the windowed kernel (library built with -DLZF_DBG_COUNT)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
d_out = torch.empty(n * BS, dtype=torch.uint8, device='cuda'); d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), 'cuda'), d_res, n); torch.cuda.synchronize()
res = device.results_to_host(d_res, n).copy()
dj = np.zeros(n, dtype=device.DJOB); d_dec = torch.empty(n * BS, dtype=torch.uint8, device='cuda')
dj['input'] = d_out.data_ptr() + np.arange(n, dtype=np.uint64) * BS
dj['input_len'] = np.where(res['status'] == 0, res['out_len'], 0)
dj['out'] = d_dec.data_ptr() + np.arange(n, dtype=np.uint64) * BS; dj['out_cap'] = BS; dj['output_limit'] = BS
d_res2 = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
device.decompress_batch(device.to_device(dj, 'cuda'), d_res2, n); torch.cuda.synchronize()
r2 = device.results_to_host(d_res2, n)
for i in range(0, n, 3):
    cl = int(dj['input_len'][i]); rv = int(r2['reserved'][i]); walks = int(r2['out_len'][i]) >> 32
    chunks = max(1, -(-cl // 65536))
    print(f"blk {i:2d} clen {cl:8d} rounds {(rv & 0xffff) * 64:8d} slow-steps {(rv >> 16) * 16:8d} walks {walks:6d} walks/chunk {walks / chunks:5.1f}")
