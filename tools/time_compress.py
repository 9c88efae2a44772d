"""Time one compress launch over `copies` x the Silesia stand-in (analysis).  usage: python tools/time_compress.py [copies] [reps]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
j1 = blocks.compress_jobs(torch.empty(1, dtype=torch.uint8, device='cuda'), BS)
if os.environ.get("LZF_ONLY_BLOCK"):     # analysis: tile one block of the corpus only (e.g. counters of a pure-text block)
    j1 = j1[int(os.environ["LZF_ONLY_BLOCK"]):][:1]; n = 1
m = n * copies
d_out = torch.empty(m * BS, dtype=torch.uint8, device='cuda')
cj = np.tile(j1, copies)
if os.environ.get("LZF_DISTINCT_BUFFERS"):      # every copy reads its own bytes (same content): separates cache locality from data effects
    d_all = d_in.repeat(copies)
    cj['input'] = (d_all.data_ptr() + (np.repeat(np.arange(copies, dtype=np.uint64), n) * np.uint64(len(data)))) + (cj['input'] - np.uint64(d_in.data_ptr()))
if os.environ.get("LZF_ORDER"):      # analysis: jobs ordered by a per-block cost list (one number per line), longest first
    cost = np.tile(np.loadtxt(os.environ["LZF_ORDER"]), copies)
    cj = cj[np.argsort(-cost, kind="stable")]
cj['out'] = d_out.data_ptr() + np.arange(m, dtype=np.uint64) * BS
d_cj = device.to_device(cj, 'cuda'); d_res = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
for it in range(reps):
    torch.cuda.synchronize(); t = time.time()
    device.compress_batch(d_cj, d_res, m); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"jobs {m} raw {len(data)*copies} time {dt*1e3:.1f} ms  {len(data)*copies/dt/2**30:.2f} GiB/s", flush=True)
res = device.results_to_host(d_res, m)
if os.environ.get("LZF_PHASES"):
    rv = res['reserved'][res['status'] == 0].astype(np.int64)
    ph = [((rv >> (8 * k)) & 255).mean() * 8.39 for k in range(4)]
    print("mean phase Mcycles per block: search %.0f extend %.0f prefetch/insert %.0f emit %.0f" % tuple(ph))
    allrv = res['reserved'].astype(np.int64).reshape(copies, n)
    for b in (0, 4, 16, 19, 28, 31, 34, 38, 41, 47, 50):
        v = allrv[:, b]; pb = [((v >> (8 * k)) & 255).mean() * 8.39 for k in range(4)]
        print("  block %2d: %4.0f %4.0f %4.0f %4.0f  (status %d, out %d)" % (b, *pb, int(res['status'][b]), int(res['out_len'][b])))
print("status counts", np.unique(res['status'], return_counts=True), "sum out_len", int(res['out_len'][res['status']==0].sum()))
