"""Avalanche / mask-statistics evaluation of candidate dropout mixers (csrc/common.h mix32 = mixA here) against lowbias32: numpy only."""
import numpy as np
M=np.uint64(0xFFFFFFFF)
def u(x): return x & M
def lowbias32(x):
    x=u(x); x^=x>>np.uint64(16); x=u(x*np.uint64(0x7feb352d)); x^=x>>np.uint64(15); x=u(x*np.uint64(0x846ca68b)); x^=x>>np.uint64(16); return x
def mad24(a,b,c): return u((a&np.uint64(0xFFFFFF))*(np.uint64(b)&np.uint64(0xFFFFFF))+c)
def mixA(x, C1=0xd3833f, C2=0x7a6b35, s1=15, s2=13, s3=16):
    x=u(x)
    x^=x>>np.uint64(s1); x=mad24(x,C1,x<<np.uint64(8) & M if False else u(x>>np.uint64(7)))
    x^=x>>np.uint64(s2); x=mad24(x,C2,u(x>>np.uint64(9)))
    x^=x>>np.uint64(s3)
    return x
def mixB(x, C1=0xd3833f, C2=0x7a6b35):
    # mad24 with the rotated self as addend keeps the top byte alive
    x=u(x)
    x^=x>>np.uint64(16); x=mad24(x,C1,u((x>>np.uint64(8))|(x<<np.uint64(24))))
    x^=x>>np.uint64(13); x=mad24(x,C2,u((x>>np.uint64(8))|(x<<np.uint64(24))))
    x^=x>>np.uint64(16)
    return x
def mixC(x, C1=0xd3833f, C2=0x7a6b35, C3=0x5bd1e9):
    x=u(x)
    x^=x>>np.uint64(16); x=mad24(x,C1,u((x>>np.uint64(8))|(x<<np.uint64(24))))
    x^=x>>np.uint64(13); x=mad24(x,C2,u((x>>np.uint64(8))|(x<<np.uint64(24))))
    x^=x>>np.uint64(15); x=mad24(x,C3,u((x>>np.uint64(8))|(x<<np.uint64(24))))
    x^=x>>np.uint64(16)
    return x
def avalanche(f, n=200000, seed=0):
    rng=np.random.RandomState(seed)
    x=rng.randint(0,2**32,size=n,dtype=np.uint64)
    h=f(x)
    worst=0; mat=np.zeros((32,32))
    for i in range(32):
        d=h^f(x^np.uint64(1<<i))
        for j in range(32):
            mat[i,j]=np.mean((d>>np.uint64(j))&np.uint64(1))
    return np.abs(mat-0.5).max(), np.abs(mat-0.5).mean()
def seq_stats(f, n=1<<20):
    # the actual use: consecutive counters x = key + i*phi (and plain consecutive), 16-bit halves vs threshold
    out={}
    for name,xs in (("phi", u(np.uint64(12345)+np.arange(n,dtype=np.uint64)*np.uint64(0x9E3779B9))), ("consec", u(np.uint64(0xabcdef01)+np.arange(n,dtype=np.uint64)))):
        h=f(xs)
        lo=(h&np.uint64(0xFFFF)).astype(np.float64); hi=(h>>np.uint64(16)).astype(np.float64)
        keep_lo=(lo>=6554); keep_hi=(hi>=6554)
        out[name]=(keep_lo.mean(), keep_hi.mean(), np.corrcoef(keep_lo,keep_hi)[0,1], np.corrcoef(keep_lo[:-1],keep_lo[1:])[0,1], np.corrcoef(keep_hi[:-1],keep_lo[1:])[0,1])
    return out
for name,f in (("lowbias32",lowbias32),("mixA",mixA),("mixB",mixB),("mixC",mixC)):
    print(name, "avalanche max/mean dev: %.3f %.4f"%avalanche(f)); 
    for k,v in seq_stats(f).items(): print("   ",k," ".join("%.4f"%t for t in v))
print("---- one round")
def mix1(x, C1=0xd3833f, s1=15, s3=16, sh=7):
    x=u(x)
    x^=x>>np.uint64(s1); x=mad24(x,C1,u(x>>np.uint64(sh)))
    x^=x>>np.uint64(s3)
    return x
for name,f in (("mix1",mix1),):
    print(name, "avalanche max/mean dev: %.3f %.4f"%avalanche(f));
    for k,v in seq_stats(f).items(): print("   ",k," ".join("%.4f"%t for t in v))
# 2-D structure test for the real use: rows = lowbias-mixed row keys, cols = pairs; look at the keep matrix autocorrelation
def grid_test(f, R=512, C=512):
    rk = lowbias32(np.arange(R,dtype=np.uint64)^np.uint64(0x1234567))
    x = u(rk[:,None] + np.arange(C,dtype=np.uint64)[None,:]*np.uint64(0x9E3779B9))
    h = f(x)
    keep = np.concatenate([((h&np.uint64(0xFFFF))>=6554)[:,:,None], ((h>>np.uint64(16))>=6554)[:,:,None]],axis=2).reshape(R,2*C).astype(np.float64)
    k = keep-keep.mean()
    res=[]
    for dr,dc in ((0,1),(0,2),(1,0),(1,1),(2,0),(0,3),(3,0),(1,2)):
        a=k[:R-dr,:2*C-dc]; b=k[dr:,dc:]
        res.append((a*b).mean()/k.var())
    return keep.mean(), np.abs(res).max()
for name,f in (("lowbias32",lowbias32),("mixA",mixA),("mix1",mix1)):
    print(name, "grid keep %.4f max |autocorr| %.4f"%grid_test(f))
