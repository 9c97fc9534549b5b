# ZSTD_compressBlock_fast (level 1, 128 KiB: hashLog 13, minMatch 6) restated in Python over one synthetic block, to
# measure what the walk touches: match offsets, age of the table candidates, share of repeat-offset matches.
# Test/analysis aid only (uses the oracle's generator); profiles/r06_zstd_enc.txt quotes its output.
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_lib import Oracle
o=Oracle()
B=131072
src=bytes(o.synth(0, 5, B, 0))
n=B
P=227718039650203
def h6(p):
    v=int.from_bytes(src[p:p+8],'little')
    return (((v<<16)&0xFFFFFFFFFFFFFFFF)*P & 0xFFFFFFFFFFFFFFFF)>>(64-13)
def r32(p): return src[p:p+4]
tab={}
ip0=1; anchor=0; off1=1; off2=0
ilimit=n-8
seqs=[]; cand_age=[]; tagmatch_age=[]
def count(a,c):
    k=0
    while a+k<n and src[a+k]==src[c+k]: k+=1
    return k
while ip0+1<ilimit:
    ip1=ip0+1; ip2=ip0+2
    h0=h6(ip0); h1=h6(ip1)
    m0=tab.get(h0,0); m1=tab.get(h1,0)
    tab[h0]=ip0; tab[h1]=ip1
    if m0: cand_age.append(ip0-m0)
    if m1: cand_age.append(ip1-m1)
    found=False
    if off1>0 and r32(ip2-off1)==r32(ip2):
        ml=1 if src[ip2-1]==src[ip2-off1-1] else 0
        ip0=ip2-ml; m=ip0-off1; ml+=4; offc=0; found=True; isrep=True
    elif m0>0 and r32(m0)==r32(ip0):
        m=m0; found=True; isrep=False
    elif m1>0 and r32(m1)==r32(ip1):
        ip0=ip1; m=m1; found=True; isrep=False
    if not found:
        st=((ip0-anchor)>>7)+2
        ip0+=st; continue
    cur0=ip1-1 if not isrep else None
    if not isrep:
        off2=off1; off1=ip0-m; ml=4
        while ip0>anchor and m>0 and src[ip0-1]==src[m-1]: ip0-=1; m-=1; ml+=1
    ml+=count(ip0+ml,m+ml)
    seqs.append((ip0-anchor, ip0-m, ml, isrep))
    start=ip0
    ip0+=ml; anchor=ip0
    # tail (approx: insertion positions)
    if ip0<=ilimit:
        # cur0+2 and ip0-2
        c0 = (start if False else None)
        tab[h6(ip0-2)]=ip0-2
        while ip0<=ilimit and off2>0 and r32(ip0)==r32(ip0-off2):
            rl=count(ip0+4,ip0+4-off2)+4
            off1,off2=off2,off1
            tab[h6(ip0)]=ip0
            seqs.append((0,off1,rl,True))
            ip0+=rl; anchor=ip0
import collections
offs=np.array([s[1] for s in seqs if not s[3]])
print("seqs",len(seqs),"rep",sum(1 for s in seqs if s[3]))
print("match offsets pct:", np.percentile(offs,[10,25,50,75,90,99]))
print("frac offsets >1024:", (offs>1024).mean(), ">608:", (offs>608).mean())
ca=np.array(cand_age); print("cand age pct", np.percentile(ca,[10,50,90]))
mls=np.array([s[2] for s in seqs]); lls=np.array([s[0] for s in seqs]); print("ml mean",mls.mean(),"ll mean",lls.mean(), "ll pct", np.percentile(lls,[50,90,99]))
