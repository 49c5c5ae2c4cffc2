"""CPU study (no GPU): how crowded level 2's 4096 equal-width depth buckets are per bin -- the in-bucket rank step of the key sort reads
sum(members^2) keys (profiles/r04_level2_counts_ab.txt, part 3).  Bins beyond the in-LDS order are cut into equal-count depth slabs here
(an estimate of k_bin_slabs' slabs).  python tools/bucket_occupancy.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg=g.load_package(); o=g.load_oracle()
def study(kind,n,w,h,shift,maxc=12288):
    rec=pkg.synth.synth_records(n,seed=0,kind=kind)
    verts=o.activate_records(rec)
    u=o.camera_uniforms(o.default_camera(),w,h)
    cov=o.cov3d(verts)
    attr,tiles=o.preprocess(verts,cov,u)
    vis=np.nonzero(tiles)[0]
    aabb=attr["aabb"][vis].astype(np.int64)
    key=attr["depth"][vis].view(np.uint32).astype(np.int64)
    S=1<<shift
    bx0=aabb[:,0]>>shift; by0=aabb[:,1]>>shift; bx1=(aabb[:,2]-1)>>shift; by1=(aabb[:,3]-1)>>shift
    tx=(w+15)//16; ty=(h+15)//16
    nbx=((tx-1)>>shift)+1; nby=((ty-1)>>shift)+1
    res=[]
    for by in range(nby):
        for bx in range(nbx):
            m=(bx0<=bx)&(bx1>=bx)&(by0<=by)&(by1>=by)
            k=key[m]
            c=len(k)
            if c<64: continue
            # slabs are not modelled: bins beyond maxc are cut into equal-count depth slabs for this estimate
            parts=[k] if c<=maxc else np.array_split(np.sort(k), -(-c//maxc))
            for kk in parts:
                kmin=kk.min(); span=kk.max()-kmin
                if span==0: continue
                bits=int(span).bit_length(); sh=max(bits-12,0)
                b=(kk-kmin)>>sh
                cnt=np.bincount(b,minlength=4096)
                used=(cnt>0).sum()
                res.append((len(kk), (cnt.astype(np.int64)**2).sum()/len(kk), cnt.max(), used))
    r=np.array(res,float)
    print(f"{kind}({n}) {w}x{h} bins of {S}: {len(r)} sorts, elements mean {r[:,0].mean():.0f}; reads per element (sum n^2 / n): mean {r[:,1].mean():.1f} median {np.median(r[:,1]):.1f} p90 {np.quantile(r[:,1],.9):.1f} max {r[:,1].max():.1f}; fullest bucket mean {r[:,2].mean():.0f} max {r[:,2].max():.0f}; buckets in use mean {r[:,3].mean():.0f} of 4096; elements/4096 = {r[:,0].mean()/4096:.1f}")
study("T",6_000_000,1920,1080,2)
study("S",6_000_000,1920,1080,2)
study("S",1_000_000,1920,1080,3,8192)
