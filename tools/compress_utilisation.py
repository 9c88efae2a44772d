"""How evenly does a compress launch fill the chip?  Sum of the jobs' wave times (results[].reserved, kilo-cycles of the job's wave)
over the wave slots the chip holds, against the launch's duration.  usage: python tools/compress_utilisation.py [copies]"""
import sys, os, time, numpy as np, torch
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
d_cj = device.to_device(cj, 'cuda'); d_res = torch.zeros(m * 16, dtype=torch.uint8, device='cuda')
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    device.compress_batch(d_cj, d_res, m); torch.cuda.synchronize()
    dt = time.time() - t
res = device.results_to_host(d_res, m)
kc = res['reserved'].astype(np.float64) * 1024.0
cu = torch.cuda.get_device_properties(0).multi_processor_count
for slots_per_cu in (17, 18):
    slots = slots_per_cu * cu
    for ghz in (2.1, 2.4):
        print(f"jobs {m}  launch {dt*1e3:.1f} ms  sum of wave times {kc.sum()/1e9:.2f} Gcycles  -> at {ghz} GHz and {slots_per_cu} waves per CU: mean busy {kc.sum()/slots/ghz/1e6:.1f} ms = {kc.sum()/slots/ghz/1e9/dt*100:.1f} % of the launch; longest job {kc.max()/ghz/1e6:.1f} ms")
