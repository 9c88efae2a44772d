"""Per-block counters of the row-mapped compress kernel (an -DLZF_DBG_ROWS build writes them over the head of each output):
loop iterations, passes through the rare phases, batches, sequences emitted on the straight path, slow batch prologues, cycles.
usage: LZF_LIB_PATH=<dbg build> python tools/rows_stats.py [copies]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
j1 = blocks.compress_jobs(torch.empty(1, dtype=torch.uint8, device='cuda'), BS)
m = n * copies
d_out = torch.zeros(m * BS, dtype=torch.uint8, device='cuda')
cj = np.tile(j1, copies)
cj['out'] = d_out.data_ptr() + np.arange(m, dtype=np.uint64) * BS
d_cj = device.to_device(cj, 'cuda'); d_res = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    device.compress_batch(d_cj, d_res, m); torch.cuda.synchronize()
    dt = time.time() - t
print(f"jobs {m} time {dt*1e3:.1f} ms  {len(data)*copies/dt/2**30:.2f} GiB/s")
res = device.results_to_host(d_res, m)
raw = d_out.view(m, BS)[:, :56].cpu().numpy().view(np.uint32).reshape(m, 14).astype(np.float64)
heads, secs = raw[:, :6], raw[:, 6:]
ok = res['status'] == 0
print("blk   iters  rare  batches direct slow  Mcyc  cyc/iter  cyc/batch  direct/batches")
for b in list(range(0, min(n, 51), 3)):
    it, rare, bat, dr, sl, cyc = heads[b]
    cyc *= 64
    print("%3d %7d %6d %7d %6d %5d %6.1f %8.0f %9.0f %6.2f  st %d" % (b, it, rare, bat, dr, sl, cyc / 1e6, cyc / max(it, 1), cyc / max(bat, 1), dr / max(bat, 1), res['status'][b]))
names = ["rare+prologue", "input wait+insert", "hash+table+tag", "cut+cand+gather", "commit+eval+bcast", "ext1", "emit+next loads", "loop top"]
for b in (0, 3, 18, 33):
    if b < m and ok[b]:
        print("block %d: cycles per batch by section (fenced timers): " % b + ", ".join("%s %.0f" % (nm, secs[b][k] * 64 / max(heads[b][2], 1)) for k, nm in enumerate(names)))
h = heads[ok]
tot = h.sum(axis=0)
print("all ok blocks: iters %.0f rare %.0f batches %.0f direct %.0f slow %.0f  cycles/iter %.0f" % (tot[0], tot[1], tot[2], tot[3], tot[4], tot[5] * 64 / tot[0]))
