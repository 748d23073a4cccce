import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle.pyoracle import Oracle
o=Oracle()
rng=np.random.default_rng(1)
n=20000; P=30
ids=np.sort(rng.choice(1<<30,size=n,replace=False)).astype(np.uint64)
e=o.roc_encode(ids,P)
head=int(e["head"]); stk=[int(w) for w in e["words"]]
L=1<<31
p1=P-16; p0=16; M1=(1<<p1)-1
bsh=P-12; sh=bsh-16
rs=np.random.RandomState(1234)
def spop():
    return stk.pop() if stk else int(rs.randint(0,2**32,dtype=np.uint64))
def upop(p):
    global head
    s=head&((1<<p)-1); head>>=p
    if head<L: head=(head<<32)|spop()
    return s
import bisect
T=[]; cnt=[0]*4096
hits=0; tot=0; spans=[]
pred=None
for i in range(n):
    upop(0); upop(0)
    xh=upop(p1); xl=upop(p0)
    x=(xh<<16)|xl
    b=x>>bsh
    if pred is not None:
        lo,span=pred
        d=(b-lo)&4095
        tot+=1; hits+= d<=span; spans.append(span)
    # rank
    r=bisect.bisect_left(T,x); bisect.insort(T,x)
    r_e=bisect.bisect_left(T,b<<bsh)  # ids in smaller buckets (T now has x, but x>=b<<bsh so unaffected)
    c=cnt[b]; cnt[b]+=1
    nmax=i+1
    h0=head
    if h0>=((L//nmax)<<32):
        stk.append(h0&0xffffffff); h0>>=32
    Hn=h0*nmax
    head=Hn+r
    v0=((Hn&0xffffffff)+r_e)&M1
    pred=((v0>>sh), min(((v0&((1<<sh)-1))+c)>>sh,7))
    if head<L:
        head=spop()|(head<<32); pred=None
print("hit rate",hits/tot,"mean rows",np.mean(spans)+1, "decoded ok", sorted(T)==list(map(int,ids)))
