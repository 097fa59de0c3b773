import numpy as np
rng=np.random.RandomState(0)
D,H,N=100,64,512
T=D*(D+1)//2
W=(0.3*rng.randn(T,H)/np.sqrt(H)).astype(np.float32)
h=np.log1p(np.exp(rng.randn(N,H))).astype(np.float32)
truth=W.astype(np.float64)@h.T.astype(np.float64)
def split16(a,scale,ftz=False):
    s=(a.astype(np.float64)*scale).astype(np.float32)
    hi=s.astype(np.float16)
    r=(s-hi.astype(np.float32))
    lo=r.astype(np.float16)
    if ftz:
        tiny=2.0**-14
        hi=np.where(np.abs(hi)<tiny,0,hi).astype(np.float16); lo=np.where(np.abs(lo)<tiny,0,lo).astype(np.float16)
    return hi.astype(np.float32),lo.astype(np.float32)
def bf16(a):
    u=a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u=((u+0x7FFF+((u>>16)&1))>>16)<<16
    return u.astype(np.uint32).view(np.float32)
def split3(a):
    a=a.astype(np.float32); h1=bf16(a); r=a-h1; m=bf16(r); l=bf16(r-m); return h1,m,l
def mm32(a,b): # fp32 accumulate emulation: float32 matmul
    return (a.astype(np.float32)@b.astype(np.float32))
ref32=mm32(W,h.T)
for ftz in (False,True):
  for sw_e,sh_e in ((16,10),(16,4),(12,0)):
    sw,sh=2.0**sw_e,2.0**sh_e
    w1,w2=split16(W,sw,ftz); h1,h2=split16(h,sh,ftz)
    acc=(mm32(w2,h1.T)+mm32(w1,h2.T))+mm32(w1,h1.T)
    got=acc.astype(np.float64)/(sw*sh)
    sab=np.abs(W.astype(np.float64))@np.abs(h.T.astype(np.float64))
    print("fp16x2 ftz",ftz,"scales",sw_e,sh_e,"max err/sum|ab|",np.abs(got-truth).max()/sab.max(), "max abs",np.abs(got-truth).max(), "maxW*s",np.abs(W).max()*sw,"maxh*s",h.max()*sh)
a=split3(W); b=split3(h)
acc=mm32(a[0],b[2].T)+mm32(a[2],b[0].T)+mm32(a[1],b[1].T)+mm32(a[0],b[1].T)+mm32(a[1],b[0].T)+mm32(a[0],b[0].T)
sab=np.abs(W.astype(np.float64))@np.abs(h.T.astype(np.float64))
print("bf16x3-6 max abs",np.abs(acc-truth).max(), "rel", np.abs(acc-truth).max()/sab.max())
print("fp32 chain max abs",np.abs(ref32-truth).max(),"rel",np.abs(ref32-truth).max()/sab.max())
# effect on x: x_k = sum_l M_kl eps_l
eps=rng.randn(N,D)
r,c=np.tril_indices(D)
def xof(Mflat):
    x=np.zeros((N,D))
    M=np.zeros((N,D,D)); M[:,r,c]=Mflat.T
    return np.einsum('nkl,nl->nk',np.tril(M,-1),eps)
xt=xof(truth)
w1,w2=split16(W,2.0**16); h1,h2=split16(h,2.0**10)
acc=((mm32(w2,h1.T)+mm32(w1,h2.T))+mm32(w1,h1.T)).astype(np.float64)/2.0**26
print("x err fp16x2:",np.abs(xof(acc)-xt).max(),"fp32:",np.abs(xof(ref32.astype(np.float64))-xt).max(), "x max",np.abs(xt).max())
# 2-term bf16 h (5 products) for comparison
h1b=bf16(h); h2b=bf16(h-h1b)
acc5=mm32(a[2],h1b.T)+mm32(a[1],h2b.T)+mm32(a[0],h2b.T)+mm32(a[1],h1b.T)+mm32(a[0],h1b.T)
print("x err bf16 five:",np.abs(xof(acc5.astype(np.float64))-xt).max())
