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
if os.environ.get("LZF_PER_BLOCK"):
    # per distinct block: compressed bytes, tokens (host walk), mean kilo-cycles of its copies (run with LZF_FED_SLOTS=0: one job per workgroup)
    r2 = device.results_to_host(d_res2, m)
    comp_h = d_out.cpu().numpy()
    print("block  C  tokens  litbytes  kcycles")
    for b in order:
        c = comp_h[int(b) * BS:int(b) * BS + int(res['out_len'][b])].tobytes(); p = 0; ntok = 0; lit = 0; L_ = len(c)
        while p < L_:
            t = c[p]; p += 1; l = t >> 4
            if l == 15:
                while True:
                    x = c[p]; p += 1; l += x
                    if x != 255: break
            p += l; lit += l; ntok += 1
            if L_ - p < 2: break
            p += 2
            if (t & 15) == 15:
                while True:
                    x = c[p]; p += 1
                    if x != 255: break
        kc = r2['reserved'][idx == b].mean()
        print(int(b), L_, ntok, lit, f"{kc:.0f}", flush=True)
if os.environ.get("LZF_TIMELINE"):
    # -DLZF_DBG_TIMELINE build: reserved = start << 16 | end on the 100 MHz wall clock, units of 2.56 us -> jobs running over time
    r2 = device.results_to_host(d_res2, m)
    st = (r2['reserved'] >> 16).astype(np.int64); en = (r2['reserved'] & 0xFFFF).astype(np.int64)
    u = np.unique(st); gaps = np.diff(np.concatenate([u, [u[0] + 65536]]))      # 16-bit circular time: the launch starts behind the largest gap
    t0 = int(u[(int(gaps.argmax()) + 1) % len(u)])
    st = (st - t0) % 65536; en = (en - t0) % 65536
    us = 2.56
    print(f"timeline: first start {st.min() * us:.0f} us, last start {st.max() * us / 1e3:.2f} ms, last end {en.max() * us / 1e3:.2f} ms, mean duration {(en - st).mean() * us / 1e3:.2f} ms")
    print(f"  durations (ms): min {(en - st).min() * us / 1e3:.2f}, median {float(np.median(en - st)) * us / 1e3:.2f}, max {(en - st).max() * us / 1e3:.2f}")
    edges = np.arange(0, en.max() + 1, max(1, int(5000 / us)))
    for e in edges:
        running = int(((st <= e) & (en > e)).sum())
        print(f"  t = {e * us / 1e3:6.1f} ms: {running:5d} jobs running, {int((st <= e).sum()):5d} started, {int((en <= e).sum()):5d} done")
if os.environ.get("LZF_SIM_SLOTS"):
    # greedy list schedule of the jobs' own wave times (longest first) on S slots: how much of the launch is packing loss?
    import heapq
    r2 = device.results_to_host(d_res2, m)
    t = np.sort(r2['reserved'].astype(np.float64))[::-1] * 1024.0
    for S in [int(x) for x in os.environ["LZF_SIM_SLOTS"].split(",")]:
        h = [0.0] * S
        for x in t:
            heapq.heappush(h, heapq.heappop(h) + x)
        print(f"slots {S}: ideal {t.sum() / S / 1e6:.1f} Mcycles, greedy makespan {max(h) / 1e6:.1f} Mcycles, longest job {t[0] / 1e6:.1f}, shortest {t[-1] / 1e6:.1f}, median {t[len(t) // 2] / 1e6:.1f}", flush=True)
if os.environ.get("LZF_VERIFY"):
    r2 = device.results_to_host(d_res2, m)
    bad = 0
    for k in range(m):
        b = int(idx[k]); ln = min(BS, len(data) - b * BS)
        if r2['status'][k] != 0 or int(r2['out_len'][k]) != ln or not torch.equal(d_dec[k * BS:k * BS + ln], d_in[b * BS:b * BS + ln]):
            bad += 1
    print(f"verify: {m - bad}/{m} jobs bit-exact", flush=True)
