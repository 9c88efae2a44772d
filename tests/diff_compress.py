"""Differential debug: first differing LZ4 sequence between GPU compress2 and the oracle."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import rust_lz_fear_amd
from rust_lz_fear_amd import ffi, synth
import oracle_ffi as o

def seqs(c):
    p=0;L=len(c);out=[];opos=0
    while p<L:
        t=c[p];p+=1;l=t>>4
        if l==15:
            while True:
                b=c[p];p+=1;l+=b
                if b!=255:break
        p+=l
        if L-p<2: out.append((opos,l,0,0)); break
        off=c[p]|(c[p+1]<<8);p+=2;m=t&15
        if m==15:
            while True:
                b=c[p];p+=1;m+=b
                if b!=255:break
        m+=4
        out.append((opos,l,m,off)); opos+=l+m
    return out

BS=4<<20
for blk in (1,2,3,4,17,25):
    data = synth.silesia_mix(blk*BS, (blk+1)*BS).tobytes()
    (rc, g), = ffi.compress_blocks_host([dict(input=data, out_cap=len(data))])
    erc, e = o.compress2(data, cap=len(data))
    if (rc, g) == (erc, e):
        print("block", blk, "equal", len(g)); continue
    sg, se = seqs(g), seqs(e)
    for i,(a,b) in enumerate(zip(sg,se)):
        if a!=b:
            print("block", blk, "first diff at seq", i, "gpu (opos,lit,mlen,off)=", a, "oracle=", b)
            print("  prev:", sg[i-1], "next gpu:", sg[i+1], "next oracle:", se[i+1])
            break
