"""Per-job time of the segmented pipeline's resolve stage (results[].reserved = the resolver's clock, k-cycles) for the corpus' blocks.
usage: LZF_LIB_PATH=<analysis lib> LZF_DECOMPRESS_KERNEL=seg LZF_SEG_MIN_IN=65536 python tools/seg_job_times.py [copies]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
d_out = torch.empty(n * BS, dtype=torch.uint8, device='cuda'); d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), 'cuda'), d_res, n); torch.cuda.synchronize()
res = device.results_to_host(d_res, n).copy()
ok = np.nonzero(res['status'] == 0)[0]
idx = np.tile(ok, copies); m = len(idx)
dj = np.zeros(m, dtype=device.DJOB); d_dec = torch.empty(m * BS, dtype=torch.uint8, device='cuda')
dj['input'] = d_out.data_ptr() + idx.astype(np.uint64) * BS; dj['input_len'] = res['out_len'][idx]
dj['out'] = d_dec.data_ptr() + np.arange(m, dtype=np.uint64) * BS; dj['out_cap'] = BS; dj['output_limit'] = BS
d_dj = device.to_device(dj, 'cuda'); d_res2 = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
for _ in range(3):
    device.decompress_batch(d_dj, d_res2, m); torch.cuda.synchronize()
r2 = device.results_to_host(d_res2, m)
if os.environ.get("LZF_TIME_SPLIT"):      # an LZF_SEG_TIME build: reserved = resolver's wait for tickets | stager fills | stager old sources | total, a byte each, 2^17 cycles
    rv = r2['reserved'].astype(np.int64)
    wait, fill, old, tot = [((rv >> sh) & 255) * 131 for sh in (0, 8, 16, 24)]      # k-cycles
    o = np.argsort(-tot)
    print("resolve stage, k-cycles per job (slowest 10 of %d; means: total %.0f, resolver waiting for tickets %.0f, stager in fills %.0f, stager on sources older than the ring %.0f)" % (m, tot.mean(), wait.mean(), fill.mean(), old.mean()))
    for k in o[:10]: print(f"  job {int(k):4d} block {int(idx[k]):3d}  total {int(tot[k]):6d}  resolver waiting {int(wait[k]):6d}  stager: fills {int(fill[k]):6d}  old sources {int(old[k]):6d}")
    sys.exit(0)
t = r2['reserved'][:len(ok)].astype(np.int64)
o = np.argsort(-t)
print("block  in_len  resolve_kcycles   (sorted by resolve time; %d blocks, %d copies)" % (len(ok), copies))
for k in o: print(f"{int(ok[k]):5d} {int(res['out_len'][ok[k]]):8d} {int(t[k]):8d}")
q = np.sort(t)[::-1]
print("quartiles of resolve time (kcycles): max %d  75%% %d  50%% %d  25%% %d  min %d" % (q[0], q[len(q)//4], q[len(q)//2], q[3*len(q)//4], q[-1]))
