"""Do two decompress launches on two streams overlap?  (a) one stream, (b) two streams with separate allocations,
(c) two streams writing slices of one allocation.  usage: python tools/stream_overlap_test.py [copies]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
def jobs_for(comp_t, dec_t, dec_off=0):
    dj = np.zeros(m, dtype=device.DJOB)
    idx = np.tile(ok, copies)
    dj['input'] = comp_t.data_ptr() + idx.astype(np.uint64) * BS
    dj['input_len'] = res['out_len'][idx]
    dj['out'] = dec_t.data_ptr() + dec_off + np.arange(m, dtype=np.uint64) * BS
    dj['out_cap'] = BS; dj['output_limit'] = BS
    return device.to_device(dj, 'cuda')
compA = d_out; compB = d_out.clone()
big = max(m * BS, 1 << 30)
decA = torch.empty(big, dtype=torch.uint8, device='cuda'); decB = torch.empty(big, dtype=torch.uint8, device='cuda')
decC = torch.empty(2 * m * BS, dtype=torch.uint8, device='cuda')
r1 = torch.zeros(m * 16, dtype=torch.uint8, device='cuda'); r2 = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
jA, jB = jobs_for(compA, decA), jobs_for(compB, decB)
jC1, jC2 = jobs_for(compA, decC), jobs_for(compA, decC, m * BS)
dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get('N_DUMMY', '0')))]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(tag, fn):
    for it in range(3):
        torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{tag}: {dt*1e3:.2f} ms", flush=True)
run("one launch (m jobs)            ", lambda: device.decompress_batch(jA, r1, m, stream=s1))
run("two launches, one stream       ", lambda: (device.decompress_batch(jA, r1, m, stream=s1), device.decompress_batch(jB, r2, m, stream=s1)))
run("two streams, separate buffers  ", lambda: (device.decompress_batch(jA, r1, m, stream=s1), device.decompress_batch(jB, r2, m, stream=s2)))
run("two streams, one output buffer ", lambda: (device.decompress_batch(jC1, r1, m, stream=s1), device.decompress_batch(jC2, r2, m, stream=s2)))

s3, s4 = torch.cuda.Stream(), torch.cuda.Stream()
jC3, jC4 = jobs_for(compB, decA), jobs_for(compB, decB)
r3 = torch.zeros(m * 16, dtype=torch.uint8, device='cuda'); r4 = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
run("four streams                   ", lambda: (device.decompress_batch(jA, r1, m, stream=s1), device.decompress_batch(jB, r2, m, stream=s2), device.decompress_batch(jC1, r3, m, stream=s3), device.decompress_batch(jC2, r4, m, stream=s4)))
