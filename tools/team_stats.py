"""Per-block counters of the team compress kernel (analysis build with -DLZF_DBG_TEAM; the counters replace the first 48 output bytes).
usage: LZF_LIB_PATH=<dbg lib> python tools/team_stats.py"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
cj = blocks.compress_jobs(torch.empty(1, dtype=torch.uint8, device='cuda'), BS)
d_out = torch.zeros(n * BS, dtype=torch.uint8, device='cuda')
cj['out'] = d_out.data_ptr() + np.arange(n, dtype=np.uint64) * BS
d_cj = device.to_device(cj, 'cuda'); d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    device.compress_batch(d_cj, d_res, n); torch.cuda.synchronize()
    print(f"call {1e3*(time.time()-t):.1f} ms")
res = device.results_to_host(d_res, n)
out = d_out.cpu().numpy().reshape(n, BS)
print("blk  st   kcyc_job |   seqs  fastB   genB | waitfill waittail gensearch   ext  total(kcyc) | more_m more_bt | cyc/seq")
for i in range(n):
    w = out[i, :80].view(np.uint32)
    print(f"{i:3d} {int(res['status'][i]):3d} {int(res['reserved'][i]):9d} | {w[0]:7d} {w[1]:7d} {w[2]:7d} | {w[3]:8d} {w[4]:8d} {w[5]:8d} {w[6]:6d} {w[9]:8d} | {w[7]:6d} {w[8]:6d} | {1024.0*w[9]/max(1,w[0]):7.0f}" + (" | sec/batch A %4.0f tag %4.0f B4 %4.0f commit %4.0f ext %4.0f push %4.0f" % tuple(1024.0 * w[10 + k] / max(1, w[1] if k < 4 else w[0]) for k in range(6)) if w[10] else ""))
