"""CPU model behind DESIGN.md section 5d: records per 2 048-rank block of a window (one bank each) on sampled windows
of the Geonames-scale haystack -- share of the read volume covered, bytes against plain entries, lanes left empty.
python tools/sim/block_records_est.py   (minutes; approximates the tokeniser: letters and one class for the rest)"""
import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tools']
import workloads as W
hay, off = W.geonames(n=2000000)
lens=(off[1:]-off[:-1]).astype(np.int64)
order=np.argsort(lens, kind='stable')
def slices(idx):
    post={}
    for rank,i in enumerate(idx):
        s=hay[int(off[i]):int(off[i+1])]
        c=np.where((s>=97)&(s<=122), s-96, np.where((s>=65)&(s<=90), s-64, 0)).astype(np.int64)
        c=np.concatenate(([0,0],c,[0]))
        t=np.unique(c[:-2]*729+c[1:-1]*27+c[2:])
        for x in t: post.setdefault(int(x),[]).append(rank)
    return post
for wstart in (500000, 1200000):
    idx=order[wstart:wstart+65520]
    post=slices(idx)
    tot_sq=sum(len(v)**2 for v in post.values())
    for MIN in (1024,2048,4096):
        raw_b=0; pk_b=0; vol_in=0; waste=0; left=0; recs=0
        for t,v in post.items():
            m=len(v)
            if m<MIN: continue
            r=np.array(v)
            nrec_b=[]; nleft=0
            for b in range(32):
                rb=r[(r>>11)==b]
                i=0; n=0
                while i<len(rb):
                    if i+15<=len(rb) and np.all(((rb[i+1:i+15]>>2)-(rb[i:i+14]>>2))<=63): n+=1; i+=15
                    else: nleft+=1; i+=1
                nrec_b.append(n)
            units=(max(nrec_b)+1)//2
            slots=units*64
            real=sum(nrec_b)
            bytes_pk=slots*16+((nleft+7)//8)*16
            w=m*m
            raw_b+=w*(2*m); pk_b+=w*bytes_pk; vol_in+=w; waste+=w*(1-real/max(1,slots)); left+=w*nleft/m
        print("window",wstart,"MIN",MIN,"share of read volume",round(vol_in/tot_sq,3),"bytes ratio (packed/raw, weighted)",round(pk_b/max(1,raw_b),3),"null-lane share",round(waste/max(1,vol_in),3),"leftover share",round(left/max(1,vol_in),3))
