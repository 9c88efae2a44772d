"""How well does a 64 KiB sample predict a block's compress time?  (analysis for the job ordering in capi.hip)
usage: python tools/probe_quality.py [copies]"""
import sys, os, heapq, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 240
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
j1 = blocks.compress_jobs(torch.empty(1, dtype=torch.uint8, device='cuda'), BS)
m = n * copies
d_out = torch.empty(m * BS, dtype=torch.uint8, device='cuda')
cj = np.tile(j1, copies)
cj['out'] = d_out.data_ptr() + np.arange(m, dtype=np.uint64) * BS
def run(jobs):
    d = device.to_device(jobs, 'cuda'); r = torch.zeros(len(jobs) * 16, dtype=torch.uint8, device='cuda')
    device.compress_batch(d, r, len(jobs)); torch.cuda.synchronize()
    return device.results_to_host(r, len(jobs)).copy()
full = run(cj)['reserved'].astype(np.float64)              # loaded per-job kilo-cycles, natural order
def makespan(t, R=4608):
    h = [0.0] * R; heapq.heapify(h)
    for x in t: heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)
ideal = full.sum() / 4608
print(f"natural {makespan(full)/ideal:.3f}  exact longest-first {makespan(np.sort(full)[::-1])/ideal:.3f}")
for S, parts in ((65536, 1), (65536, 4), (32768, 1), (131072, 1), (131072, 4)):
    pj = cj.copy()
    payload = pj['input_len'] - pj['cursor']
    big = payload >= 4 * S
    est = np.zeros(m)
    for k in range(parts):
        q = pj.copy()
        piece = S // parts
        off = ((payload - piece) * (2 * k + 1) // (2 * parts)) & ~np.uint64(15)
        q['input'] = pj['input'] + pj['cursor'] + off
        q['input_len'] = np.where(big, piece, 0); q['cursor'] = 0; q['out_cap'] = 2 * piece
        est += run(q)['reserved'].astype(np.float64)
    est = np.where(big, est * payload / S, payload * 0.2)
    order = np.argsort(-est, kind='stable')
    cc = np.corrcoef(est[big], full[big])[0, 1]
    print(f"sample {S} in {parts} part(s): corr {cc:.3f}  longest-first by estimate {makespan(full[order])/ideal:.3f}")
