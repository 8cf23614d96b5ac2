"""CPU model of feistel_small (csrc/common.hpp) and of the round-key derivations of select.hip: max |share - p| / sigma of the n_sel-subset draw
over T seeds, for small-integer and hash-like seeds (how the round-4 key fix was chosen).  Experiments only."""
import numpy as np
np.seterr(over='ignore')
U=np.uint32
def mix32(x):
    x=x.astype(U); x^=x>>U(16); x*=U(0x7feb352d); x^=x>>U(15); x*=U(0x846ca68b); x^=x>>U(16); return x
def feistel(j,n,bits,k0,k1,variant,rounds=4):
    rb=bits>>1; lb=bits-rb; rm=U((1<<rb)-1); lm=U((1<<lb)-1)
    x=j.astype(U).copy(); done=np.zeros(x.shape,bool); out=np.zeros_like(x)
    while not done.all():
        l=x>>U(rb); r=x&rm
        for rd in range(rounds):
            half = l if rd&1 else r
            key = k1 if rd&1 else k0
            f=(half*U(0x9E3779B1)+key+U((rd*0x7F4A7C15)&0xffffffff)).astype(U)
            if variant==0:
                f^=f>>U(15); f*=U(0x846ca68b); f^=f>>U(13)
            else:
                f=mix32(f); f=f>>U(9)
            if rd&1: r=r^(f&rm)
            else: l=l^(f&lm)
        x=((l<<U(rb))|r).astype(U)
        newly=(~done)&(x<n)
        out[newly]=x[newly]; done|=newly
    return out
def bits_for(n):
    b=2
    while (1<<b)<n: b+=1
    return b
def test(k,n_sel,T,variant,rounds,keymix,seedfn):
    cnt=np.zeros(k)
    for s in range(T):
        seed=seedfn(s)
        lo=U(seed&0xffffffff); hi=U(seed>>32)
        k0=mix32(np.array([lo^U(0x9E3779B9)],dtype=U))[0]
        if keymix: k1=mix32(np.array([hi+U(0x85EBCA6B)+k0*U(0x632BE5AB)],dtype=U))[0]
        else: k1=mix32(np.array([hi+U(0x85EBCA6B)],dtype=U))[0]
        j=feistel(np.arange(n_sel,dtype=U),k,bits_for(k),k0,k1,variant,rounds)
        assert len(set(j.tolist()))==n_sel
        cnt[j]+=1
    p=n_sel/k; f=cnt/T; sig=np.sqrt(p*(1-p)/T)
    return (np.abs(f-p).max()/sig)
for name,seedfn in (("small", lambda s:7919*s+3),("hash", lambda s:(s*0xD1B54A32D192ED03+0x1234567)&0xffffffffffffffff)):
  for (k,n_sel) in ((56,14),(24,12),(48,24),(300,150),(7,3)):
    r=[]
    for variant,rounds,keymix in ((0,4,0),(0,4,1),(1,4,1),(1,6,1),(0,6,1)):
        r.append("%.1f"%test(k,n_sel,2000,variant,rounds,keymix,seedfn))
    print(name,k,n_sel,"max|dev|/sigma: cur4=%s cur4+key=%s mix4=%s mix6=%s cur6=%s"%tuple(r))
