# LZ4_compress_fast (liblz4 1.9.3, byU32, hash log 12, acceleration 1) restated in Python over one synthetic block, to measure
# what the search touches: match offsets, age of the table candidates, how many candidates are older than the encoder's LDS ring.
# Analysis aid only (uses the oracle's generator); profiles/r06_lz4_enc.txt quotes its output.
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_lib import Oracle
o = Oracle()
B = 131072
dist = int(sys.argv[1]) if len(sys.argv) > 1 else 0
src = bytes(o.synth(0, 5, B, dist))
n = B
P = 889523592379
def h5(p):
    v = int.from_bytes(src[p:p + 8], 'little')
    return (((v << 24) & 0xFFFFFFFFFFFFFFFF) * P & 0xFFFFFFFFFFFFFFFF) >> 52
tab = {}
mflimit_p1 = n - 12 + 1
matchlimit = n - 5
ip = 0; anchor = 0
tab[h5(0)] = 0
ip = 1
seqs = []; ages = []; probes_per_seq = []; far_before_hit = []
RING = 1024   # bytes of back window the 2 KiB ring guarantees
done = False
fwdh = h5(ip)
while not done:
    # search
    fwd = ip; step = 1; nb = 1 << 6; probes = 0; far = 0
    while True:
        h = fwdh; ip = fwd; fwd += step; step = nb >> 6; nb += 1
        if fwd > mflimit_p1: done = True; break
        m = tab.get(h, 0)
        fwdh = h5(fwd)
        tab[h] = ip
        probes += 1
        age = ip - m
        ages.append(age)
        if age > RING and m + 65535 >= ip: far += 1
        if m + 65535 < ip: continue
        if src[m:m + 4] == src[ip:ip + 4]: break
    if done: break
    probes_per_seq.append(probes); far_before_hit.append(far)
    while ip > anchor and m > 0 and src[ip - 1] == src[m - 1]: ip -= 1; m -= 1
    lit = ip - anchor
    a = ip + 4; b = m + 4
    while a < matchlimit and src[a] == src[b]: a += 1; b += 1
    ml = a - ip - 4
    seqs.append((lit, ip - m, ml + 4))
    ip = a; anchor = ip
    if ip >= mflimit_p1: break
    tab[h5(ip - 2)] = ip - 2
    # test next position (the serial code re-tests ip immediately; modelled as the first probe of the next search)
    fwdh = h5(ip)
offs = np.array([s[1] for s in seqs]); ages = np.array(ages)
print("dist", dist, "sequences", len(seqs), "probes/seq %.1f" % np.mean(probes_per_seq))
print("match offsets pct 10/25/50/75/90/99:", np.percentile(offs, [10, 25, 50, 75, 90, 99]), "frac > %d: %.3f" % (RING, (offs > RING).mean()))
print("candidate age pct 10/50/90:", np.percentile(ages, [10, 50, 90]), "frac > %d: %.3f" % (RING, (ages > RING).mean()))
print("far candidates probed before the hit, per sequence: mean %.1f" % np.mean(far_before_hit))
print("match length mean %.1f, literal run mean %.1f" % (np.mean([s[2] for s in seqs]), np.mean([s[0] for s in seqs])))
