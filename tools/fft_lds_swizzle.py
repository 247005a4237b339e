#!/usr/bin/env python3
"""Developer tool: LDS bank conflicts of FbankKernel's 256-point split-radix FFT (frontend.hip) per frame, for the plain
layout and for every XOR swizzle slot(i) = i ^ T(i >> 4) (T a 4 x 4 matrix over GF(2)); prints the best.
Model: 32 banks x 4 B, a 64-lane 8-byte access served 16 lanes per cycle, equal addresses broadcast."""
import itertools, sys
def enumerate_blocks(off, logm, out):
    if logm <= 0: return
    out.append((off, logm))
    if logm >= 2:
        m = 1 << logm
        enumerate_blocks(off, logm-1, out)
        enumerate_blocks(off + m//2, logm-2, out)
        enumerate_blocks(off + 3*(m//4), logm-2, out)
blocks=[]; enumerate_blocks(0,8,blocks)
blocks.sort(key=lambda b:(-b[1], b[0]))
passes={}
for off,logm in blocks: passes.setdefault(logm,[]).append(off)
def bitrev(i):
    r=0
    for _ in range(8): r=(r<<1)|(i&1); i>>=1
    return r
# access list: each access = list of 64 indices (or None)
acc=[]
for r in range(4): acc.append(('stage',[l+64*r for l in range(64)]))
for logm in range(8,1,-1):
    q=(1<<logm)//4; bl=passes[logm]
    for pt in range(4):
        idx=[]
        for lane in range(64):
            b,n=divmod(lane,q)
            idx.append(bl[b]+n+pt*q if b<len(bl) else None)
        acc.append(('p%d r'%logm, idx)); acc.append(('p%d w'%logm, idx))
bl=passes[1]
for r in range(2):
    for pt in range(2):
        idx=[(bl[l+64*r]+pt) if l+64*r<len(bl) else None for l in range(64)]
        acc.append(('2pt r',idx)); acc.append(('2pt w',idx))
for r in range(2):
    acc.append(('post k',[bitrev(1+l+64*r) for l in range(64)]))
    acc.append(('post d',[bitrev(256-(1+l+64*r)) for l in range(64)]))
def cost(phys, detail=False):
    tot=0; per={}
    for name,idx in acc:
        c=0
        for g in range(4):
            cnt={}
            seen=set()
            for lane in range(16*g,16*g+16):
                i=idx[lane]
                if i is None: continue
                p=phys[i]
                if p in seen: continue      # same address: broadcast
                seen.add(p)
                cnt[p%16]=cnt.get(p%16,0)+1
            c+= (max(cnt.values())-1) if cnt else 0
        tot+=c; per[name]=per.get(name,0)+c
    return (tot,per) if detail else tot
ident=list(range(256))
print('identity', cost(ident,True))
best=None
for T in range(65536):
    cols=[(T>>(4*j))&15 for j in range(4)]   # image of high bit j
    phys=[]
    for i in range(256):
        h=i>>4; x=0
        for j in range(4):
            if (h>>j)&1: x^=cols[j]
        phys.append((i&~15)|((i&15)^x))
    c=cost(phys)
    if best is None or c<best[0]: best=(c,T,cols); 
print('best', best)
cols=best[2]
phys=[]
for i in range(256):
    h=i>>4; x=0
    for j in range(4):
        if (h>>j)&1: x^=cols[j]
    phys.append((i&~15)|((i&15)^x))
print(cost(phys,True))
