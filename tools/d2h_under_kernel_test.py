"""Does a device -> pinned host copy slow down while a decompress launch (the segmented pipeline: one workgroup of 128 KiB LDS
per CU for ~7 ms) runs on another stream?  usage: python tools/d2h_under_kernel_test.py [copies]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 3
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
d_out = torch.empty(n * BS, dtype=torch.uint8, device='cuda')
d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), 'cuda'), d_res, n); torch.cuda.synchronize()
res = device.results_to_host(d_res, n).copy()
ok = np.nonzero(res['status'] == 0)[0]
m = len(ok) * copies
dec = torch.empty(m * BS, dtype=torch.uint8, device='cuda')
dj = np.zeros(m, dtype=device.DJOB); idx = np.tile(ok, copies)
dj['input'] = d_out.data_ptr() + idx.astype(np.uint64) * BS; dj['input_len'] = res['out_len'][idx]
dj['out'] = dec.data_ptr() + np.arange(m, dtype=np.uint64) * BS; dj['out_cap'] = BS; dj['output_limit'] = BS
jobs = device.to_device(dj, 'cuda'); r = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
NB = 512 << 20; PIECE = 4 << 20
src = torch.empty(NB, dtype=torch.uint8, device='cuda'); pin = torch.empty(NB, dtype=torch.uint8).pin_memory()
hsrc = torch.empty(NB, dtype=torch.uint8).pin_memory(); dst = torch.empty(NB, dtype=torch.uint8, device='cuda')
sk, c1, c2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def d2h(nstreams=2):
    for k in range(NB // PIECE):
        with torch.cuda.stream(c1 if (k & 1) == 0 or nstreams == 1 else c2): pin[k * PIECE:(k + 1) * PIECE].copy_(src[k * PIECE:(k + 1) * PIECE], non_blocking=True)
def h2d():
    for k in range(NB // PIECE):
        with torch.cuda.stream(c1 if (k & 1) == 0 else c2): dst[k * PIECE:(k + 1) * PIECE].copy_(hsrc[k * PIECE:(k + 1) * PIECE], non_blocking=True)
def timed(tag, fn, what_ms=None):
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"{tag}: {best*1e3:.2f} ms", flush=True); return best
tk = timed(f"decompress {m} blocks alone      ", lambda: device.decompress_batch(jobs, r, m, stream=sk))
tc = timed("D2H 512 MiB alone (2 streams)     ", d2h); print(f"   {NB/tc/1e9:.1f} GB/s")
tc1 = timed("D2H 512 MiB alone (1 stream)      ", lambda: d2h(1)); print(f"   {NB/tc1/1e9:.1f} GB/s")
th = timed("H2D 512 MiB alone (2 streams)     ", h2d); print(f"   {NB/th/1e9:.1f} GB/s")
timed("decompress + D2H together         ", lambda: (device.decompress_batch(jobs, r, m, stream=sk), d2h()))
timed("decompress + H2D together         ", lambda: (device.decompress_batch(jobs, r, m, stream=sk), h2d()))
timed("D2H + H2D together                ", lambda: (d2h(), h2d()))

# how long the copies themselves take next to a running launch (events on the copy streams)
def copy_time_with_kernel(copy_fn, tag):
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        device.decompress_batch(jobs, r, m, stream=sk)
        for s_, a in ((c1, 0), (c2, 1)): e0[a].record(s_)
        copy_fn()
        for s_, a in ((c1, 0), (c2, 1)): e1[a].record(s_)
        torch.cuda.synchronize()
        best = min(best, max(e0[a].elapsed_time(e1[a]) for a in range(2)))
    print(f"{tag}: {best:.2f} ms  ({NB / best / 1e6:.1f} GB/s)", flush=True)
copy_time_with_kernel(d2h, "D2H 512 MiB next to the launch     ")
copy_time_with_kernel(h2d, "H2D 512 MiB next to the launch     ")
