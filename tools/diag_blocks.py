"""Per-block kernel cycles (and phase timers with a -DLZF_PHASE_TIMING build + LZF_PHASES=1) over the Silesia stand-in."""
import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo') if os.path.isdir('/root/repo') else None
sys.path.insert(0, os.getcwd())
import rust_lz_fear_amd
from rust_lz_fear_amd import device, ffi, synth
BS=4<<20
data = synth.silesia_mix()
d_in = torch.from_numpy(data).cuda()
blocks = device.BlockSet(d_in, BS)
n=blocks.n
d_out = torch.empty(n*BS, dtype=torch.uint8, device='cuda')
d_jobs = device.to_device(blocks.compress_jobs(d_out, BS), 'cuda')
d_res = torch.zeros(n*16, dtype=torch.uint8, device='cuda')
device.compress_batch(d_jobs, d_res, n); torch.cuda.synchronize()
res = device.results_to_host(d_res, n).copy()
dj = np.zeros(n, dtype=device.DJOB)
d_dec = torch.zeros(n*BS, dtype=torch.uint8, device='cuda')
dj['input'] = d_out.data_ptr() + np.arange(n, dtype=np.uint64)*BS
dj['input_len'] = np.where(res['status']==0, res['out_len'], 0)
dj['out'] = d_dec.data_ptr() + np.arange(n, dtype=np.uint64)*BS
dj['out_cap']=BS; dj['output_limit']=BS
d_dj = device.to_device(dj,'cuda'); d_res2 = torch.zeros(n*16, dtype=torch.uint8, device='cuda')
for it in range(2):
    device.decompress_batch(d_dj, d_res2, n); torch.cuda.synchronize()
r2 = device.results_to_host(d_res2, n)
# segment name per block
names=[]; pos=0; bounds=[]
for nm,ln,cls,kw in synth.SILESIA_SEGMENTS:
    bounds.append((pos,pos+ln,nm)); pos+=ln
def seg(b):
    mid=b*BS+BS//2
    for a,e,nm in bounds:
        if a<=mid<e: return nm
    return '?'
# sequences per block via a CPU token walk on the compressed bytes
comp = d_out.cpu().numpy()
def nseq(c):
    p=0;k=0;L=len(c)
    while p<L:
        t=c[p];p+=1;l=t>>4
        if l==15:
            while True:
                b=c[p];p+=1;l+=b
                if b!=255:break
        p+=l
        if L-p<2:break
        p+=2;m=t&15
        if m==15:
            while True:
                b=c[p];p+=1
                if b!=255:break
        k+=1
    return k
PH = os.environ.get('LZF_PHASES')
print('blk seg       clen   comp_kcyc  dec_kcyc  status  nseq')
for i in range(n):
    cl=int(res['out_len'][i]) if res['status'][i]==0 else 0
    ns = nseq(comp[i*BS:i*BS+cl].tolist()) if (cl and i%4==0) else -1
    if PH:
        cp = int(res['reserved'][i]); cph = [((cp >> (8*k)) & 255) * 8.4 for k in range(4)]
        print(f"{i:3d} {seg(i):9s} compress phases(Mcyc) search={cph[0]:.0f} extend={cph[1]:.0f} insert/prefetch={cph[2]:.0f} emit={cph[3]:.0f}")
        pk = (int(r2['out_len'][i]) >> 32) | (int(r2['reserved'][i]) << 32)
        ph = [(pk >> (10*k)) & 1023 for k in range(6)]
        print(f"{i:3d} {seg(i):9s} {cl:8d} phases(Mcyc) parse={ph[0]} setup/err={ph[1]} lit={ph[2]} far={ph[3]} rounds={ph[4]} flush={ph[5]} nseq={ns}")
    else:
        print(f"{i:3d} {seg(i):9s} {cl:8d} {int(res['reserved'][i]):9d} {int(r2['reserved'][i]):9d} {int(res['status'][i])} {ns}")
