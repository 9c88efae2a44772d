"""One decompress launch over `copies` x the Silesia stand-in's blocks (analysis: PMC / timing of kernel variants).
usage: LZF_LIB_PATH=... python tools/pmc_decomp.py [copies] [reps]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BS = int(os.environ.get("LZF_BS", 4 << 20))
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS)
n = blocks.n
d_out = torch.empty(n * BS, dtype=torch.uint8, device='cuda')
d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), 'cuda'), d_res, n); torch.cuda.synchronize()
res = device.results_to_host(d_res, n).copy()
ok = np.nonzero(res['status'] == 0)[0]
order = ok[np.argsort(-res['out_len'][ok].astype(np.int64))]
m = len(order) * copies
dj = np.zeros(m, dtype=device.DJOB)
d_dec = torch.empty(m * BS, dtype=torch.uint8, device='cuda')
idx = np.repeat(order, copies) if not os.environ.get('LZF_NOSORT') else np.tile(np.sort(order), copies)   # LZF_NOSORT: corpus order, tiled
dj['input'] = d_out.data_ptr() + idx.astype(np.uint64) * BS
dj['input_len'] = res['out_len'][idx]
dj['out'] = d_dec.data_ptr() + np.arange(m, dtype=np.uint64) * BS
dj['out_cap'] = BS; dj['output_limit'] = BS
d_dj = device.to_device(dj, 'cuda'); d_res2 = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
raw = int(sum(min(BS, len(data) - int(b) * BS) for b in order)) * copies
for it in range(reps):
    torch.cuda.synchronize(); t = time.time()
    device.decompress_batch(d_dj, d_res2, m); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"jobs {m} raw {raw} time {dt*1e3:.2f} ms  {raw/dt/2**30:.1f} GiB/s", flush=True)
if os.environ.get("LZF_PRINT_RESERVED"):
    r2 = device.results_to_host(d_res2, m)
    print(f"reserved: mean {r2['reserved'].mean():.1f} kcycles/job, sum {int(r2['reserved'].sum())} ; status ok {int((r2['status'] == 0).sum())}/{m}", flush=True)
if os.environ.get("LZF_VERIFY"):
    r2 = device.results_to_host(d_res2, m)
    bad = 0
    for k in range(m):
        b = int(idx[k]); ln = min(BS, len(data) - b * BS)
        if r2['status'][k] != 0 or int(r2['out_len'][k]) != ln or not torch.equal(d_dec[k * BS:k * BS + ln], d_in[b * BS:b * BS + ln]):
            bad += 1
    print(f"verify: {m - bad}/{m} jobs bit-exact", flush=True)
