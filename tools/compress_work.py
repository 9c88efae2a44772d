"""Probe batches and sequences per block (library built with -DLZF_DBG_DRY_MAIN: LZF_LIB_PATH=dbg/lib_dry.so) next to the
block's standalone kernel time (default library) — analysis of where the compress kernel's time goes."""
import sys, os, subprocess, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, synth
BS = 4 << 20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS); n = blocks.n
d_out = torch.empty(n * BS, dtype=torch.uint8, device='cuda'); d_res = torch.zeros(n * 16, dtype=torch.uint8, device='cuda')
device.compress_batch(device.to_device(blocks.compress_jobs(d_out, BS), 'cuda'), d_res, n); torch.cuda.synchronize()
res = device.results_to_host(d_res, n).copy()
print(json.dumps([int(x) for x in res['reserved']]))
