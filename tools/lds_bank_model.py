#!/usr/bin/env python
"""LDS bank-conflict model of ds_read_b128 / ds_write_b128 (lane groups and bank widths of MI355X_MICROARCH.md, section LDS) for the attention kernels' K and V^T
images, and a brute-force search for V^T key-block permutations that are conflict-free for the MFMA fragment reads (profiles/r06_c_attention_pipes.md).
usage: python tools/lds_bank_model.py"""
import itertools
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        [x+32 for x in list(range(0,4))+list(range(12,16))+list(range(20,28))], [x+32 for x in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
def cycles(addr_of_lane, groups, width_dw, nbanks):
    tot=0
    for g in groups:
        banks={}
        for l in g:
            a=addr_of_lane(l)
            if a is None: continue
            for k in range(width_dw):
                dw=a//4+k
                banks.setdefault(dw%nbanks,set()).add(dw)
        tot+=max([len(v) for v in banks.values()] or [1])
    return tot
# new V^T layout reads
def vt200(kb0, dt):
    def f(l):
        l31=l&31; hf=l>>5; d=dt*32+l31
        pos=(kb0+hf+((d>>4)&3))%25
        return d*400+pos*16
    return f
def vt264(kb0, dt):
    def f(l):
        l31=l&31; hf=l>>5; d=dt*32+l31
        kb=kb0+hf
        return d*528+((kb^((d>>3)&7))<<4)
    return f
for name,fn in (("vt200",vt200),("vt264",vt264)):
    worst=0;tot=0;n=0
    for kb0 in range(0,28,2):
        for dt in (0,1):
            c=cycles(fn(kb0,dt),G128,4,64); worst=max(worst,c); tot+=c;n+=1
    print(name,"read cycles avg",tot/n,"worst",worst,"(ideal 4)")
# writes: ds_write_b128, 8 groups of 8 contiguous lanes, banks mod 32 (per the table) -- thread (kb,c)= (tid>>3, tid&7), row d=c*8+j
GW=[list(range(i*8,i*8+8)) for i in range(8)]
def w200(wave,j):
    def f(l):
        tid=wave*64+l
        if tid>=200: return None
        kb=tid>>3;c=tid&7;d=c*8+j
        pos=(kb+(c>>1))%25
        return d*400+pos*16
    return f
def w264(wave,j):
    def f(l):
        tid=wave*64+l
        if tid>=224: return None
        kb=tid>>3;c=tid&7;d=c*8+j
        return d*528+((kb^c)<<4)
    return f
for name,fn in (("w200",w200),("w264",w264)):
    for nb in (32,64):
        tot=0;n=0;worst=0
        for wave in range(4):
            for j in range(8):
                c=cycles(fn(wave,j),GW,4,nb);tot+=c;n+=1;worst=max(worst,c)
        print(name,"banks",nb,"write cycles avg",tot/n,"worst",worst,"(ideal 8)")
def kread(t,ks):
    def f(l):
        l31=l&31; hf=l>>5; row=t*32+l31; ch=ks*2+hf
        return ((row>>1)<<8)+((((row&1)<<3)|ch)^((row>>1)&15))*16
    return f
tot=0;n=0
for t in range(7):
    for ks in range(4):
        tot+=cycles(kread(t,ks),G128,4,64);n+=1
print("K frag read cycles avg",tot/n)
# K writes (ds_write_b128): thread idx -> row idx>>3, chunk idx&7
def kwrite(wave,it,NT):
    def f(l):
        idx=wave*64+l+it*NT; r=idx>>3;c=idx&7
        return ((r>>1)<<8)+((((r&1)<<3)|c)^((r>>1)&15))*16
    return f
print("K write cycles", cycles(kwrite(0,0,512),GW,4,32), cycles(kwrite(0,0,512),GW,4,64))
# candidate V^T layouts
def vtS(kb0,dt,s,g):
    def f(l):
        l31=l&31; hf=l>>5; d=dt*32+l31
        pos=(kb0+hf+g[d>>3])%s
        return d*s*16+pos*16
    return f
for s,g in ((25,(0,0,0,0,1,1,1,1)),(26,(0,0,1,1,2,2,3,3)),(28,tuple(range(8)))):
    tot=0;n=0
    for kb0 in range(0,28,2):
        for dt in (0,1):
            tot+=cycles(vtS(kb0,dt,s,g),G128,4,64);n+=1
    print("V^T s",s,"read cycles avg",tot/n)
import itertools
A=[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]; B=[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
def read_ok(s,g):
    # all kb0 (even), hf adds 1 -> separate lane groups (hf=1 lanes are own groups) ; dt in 0,1
    for grp in (A,B):
        for dt in (0,1):
            slots=set()
            for l31 in grp:
                d=dt*32+l31
                slots.add((d*s+g[d>>3])%16)
            if len(slots)<16: return False
    return True
def write_cost(s,g,nb_slots):
    # 8 lanes c=0..7 ; slot = ((8c+j)*s + g[c]) mod nb_slots ; return max multiplicity
    worst=0
    for j in range(8):
        sl={}
        for c in range(8):
            k=((8*c+j)*s+g[c])%nb_slots
            sl[k]=sl.get(k,0)+1
        worst=max(worst,max(sl.values()))
    return worst
for s in range(25,34):
    best=None
    for g in itertools.product(range(8),repeat=8):
        if g[0]!=0: continue
        if read_ok(s,g):
            w8=write_cost(s,g,8); w16=write_cost(s,g,16)
            key=(w8,w16,max(g))
            if best is None or key<best[0]: best=(key,g)
    print(s,best)
