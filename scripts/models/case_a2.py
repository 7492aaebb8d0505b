"""Python model of lane_case_a2_jump and lane_case_a2 (return run + "other" underfull, listed slots overfull) in node2vec_amd/csrc/n2v_unit_core.h, checked against the pairing loop
of generate_alias_tables (reference randomwalk.py:175-189, restated in ref_tables) on random rows of
the three class values: `python case_a2.py` (short rows), `python case_a2.py big` (long rows); N2V_MODEL_TRIALS
overrides the number of rows.  Prints the mismatch count (must be 0) and how many rows the closed
form leaves to the replay ("ambiguous")."""
import os
import random, math, struct, sys
def ref_tables(w):
    n=len(w); alias=[0]*n; avg=sum(w)/n; probs=[x/avg for x in w]
    under=[i for i in range(n) if probs[i]<1.0]; over=[i for i in range(n) if not probs[i]<1.0]
    while under and over:
        u,o=under.pop(),over.pop(); alias[u]=o; probs[o]=probs[o]+probs[u]-1.0
        (under if probs[o]<1.0 else over).append(o)
    return alias,probs
def bexp(x): return (struct.unpack('<q',struct.pack('<d',x))[0]>>52)&0x7ff
def absorb_skip(r,val,j,limit):
    while j+3<limit:
        t1=r+val; r1=t1-1.0; t2=r1+val; r2=t2-1.0; t3=r2+val; r3=t3-1.0
        if r3<1.0: return r,j
        e1=bexp(t1); e3=bexp(t3)
        j+=3; r=r3
        if e1!=e3: continue
        edge=struct.unpack('<d',struct.pack('<q',e3<<52))[0]
        d=r2-r3
        room=limit-j
        if d==0.0: j+=room; return r,j
        if not (t3>edge): continue
        nn=math.floor((t3-edge)/d)-1.0
        nn=min(nn,float(room))
        while nn>=1.0 and not (t3-nn*d>edge): nn-=1.0
        if nn>=1.0: r=r3-nn*d; j+=int(nn)
        if nn<4.0: return r,j
    return r,j

def jump_a2(n,cls,pick,r2,gR,gM,gO):
    Rpos=[i for i in range(n) if cls[i]=='R']; lst=[i for i in range(n) if cls[i]=='M']
    nR,nM=len(Rpos),len(lst); nO=n-nR-nM; rpos=Rpos[0] if Rpos else 0
    isum=nR*gR+nM*gM+nO*gO; EM=gM*n-isum; D=isum-gO*n; DR=isum-gR*n
    if not (D>0 and DR>0 and nM>0 and EM>0 and nR>0): return None
    lo_r=sum(1 for x in lst if x<rpos); mA=nM-lo_r
    rho=(n-rpos-nR)-mA; nU=nO+nR
    def Def(k):
        if k<=rho: return k*D
        if k<=rho+nR: return rho*D+(k-rho)*DR
        return rho*D+nR*DR+(k-rho-nR)*D
    lo_pick=sum(1 for x in lst if x<pick)
    if cls[pick]!='M':
        if cls[pick]=='R': k=rho+(rpos+nR-1-pick)
        else: k=(n-1-pick)-(nM-lo_pick)
        T=Def(k)
        if T<=0: i=1
        else:
            i=(T+EM-1)//EM
            if i*EM==T: return None
        if i<1 or i>nM: return None
        return lst[nM-i]
    i0=nM-lo_pick
    if i0<1 or i0>nM: return None
    if i0==nM: return pick
    X=i0*EM
    if rho*D>X: k=X//D+1
    else:
        Y1=rho*D
        if Y1+nR*DR>X: k=rho+(X-Y1)//DR+1
        else: k=rho+nR+(X-Y1-nR*DR)//D+1
    if k<1 or k>nU: return None
    if Def(k-1)==X: return None
    prob=1.0+(X-Def(k))/isum
    if abs(prob-r2)<1e-9: return None
    return pick if r2<prob else lst[nM-(i0+1)]

def lane_case_a2(n,cls,pick,r2,vR,vM,vO):
    Rpos=[i for i in range(n) if cls[i]=='R']; lst=[i for i in range(n) if cls[i]=='M']
    nR,nM=len(Rpos),len(lst); nO=n-nR-nM; rpos=Rpos[0] if Rpos else 0
    pickR=cls[pick]=='R'; pickM=cls[pick]=='M'
    nU=nO+nR
    lo_r=sum(1 for x in lst if x<rpos); mA=nM-lo_r
    rho=(n-rpos-nR)-mA
    rank=-1
    if pickR: rank=rho+(rpos+nR-1-pick)
    elif not pickM:
        above=sum(1 for x in lst if x>pick)
        rank=(n-1-pick)-above
    km=nM-1; used=0; have_carry=False; carry_i=0; alias_pick=0; carry_v=0.0
    p_pick={'R':vR,'M':vM,'O':vO}[cls[pick]]
    while True:
        if not have_carry and used>=nU: break
        if km<0: break
        oi=lst[km]; ov=vM; km-=1
        if have_carry:
            if carry_i==pick: alias_pick=oi; p_pick=carry_v; break
            ov=ov+carry_v-1.0; have_carry=False
            if ov<1.0:
                if oi==pick: p_pick=ov
                have_carry=True; carry_i=oi; carry_v=ov; continue
        demoted=False
        while True:
            if used>=nU: break
            if rank>=0 and used==rank: break
            if used<rho: val=vO; seg_end=rho
            elif used<rho+nR: val=vR; seg_end=rho+nR
            else: val=vO; seg_end=nU
            limit=seg_end-used
            if rank>=used and rank-used<limit: limit=rank-used
            j=0
            ov,j=absorb_skip(ov,val,j,limit)
            while j<limit:
                ov=ov+val-1.0; j+=1
                if ov<1.0: demoted=True; break
            used+=j
            if demoted: break
        if oi==pick: p_pick=ov
        if demoted:
            have_carry=True; carry_i=oi; carry_v=ov; continue
        if rank>=0 and used==rank and used<nU:
            alias_pick=oi; p_pick=(vR if (rank>=rho and rank<rho+nR) else vO)
        break
    return pick if r2<p_pick else alias_pick

random.seed(7); bad=0; amb=0; tot=0; bad2=0
big=len(sys.argv)>1
for trial in range(int(os.environ.get("N2V_MODEL_TRIALS", 5000 if big else 200000))):
    n=random.choice([100,300,1000,3000]) if big else random.randint(2,70)
    p,q=random.choice([(4.0,2.0),(2.0,2.0**0.5) if False else (4.0,2.0),(8.0,2.0),(8.0,4.0),(16.0,2.0)])
    bR,bM,bO=1/p,1.0,1/q
    TR,TM,TO=int(bR*2**20),int(bM*2**20),int(bO*2**20); g=math.gcd(TR,math.gcd(TM,TO))
    cls=['O']*n
    nR=min(n,random.choice([1,1,1,2,3])); rp=random.randint(0,n-nR)
    for k in range(nR): cls[rp+k]='R'
    fr=random.choice([0.02,0.1,0.3,0.6])
    for i in range(n):
        if cls[i]=='O' and random.random()<fr: cls[i]='M'
    w=[{'R':bR,'M':bM,'O':bO}[c] for c in cls]
    avg=sum(w)/n; v={'R':bR/avg,'M':bM/avg,'O':bO/avg}
    cnt={c:cls.count(c) for c in 'RMO'}
    if not (v['O']<1 and cnt['R'] and v['R']<1 and cnt['M'] and not v['M']<1): continue
    alias,probs=ref_tables(w)
    pick=random.randrange(n); r2=random.getrandbits(32)/2**32
    p0=v[cls[pick]]
    if p0<1.0 and r2<p0: continue
    want = pick if r2<probs[pick] else alias[pick]
    got=jump_a2(n,cls,pick,r2,TR//g,TM//g,TO//g)
    tot+=1
    if got is None: amb+=1
    elif got!=want:
        bad+=1
        if bad<6: print("MISMATCH jump",n,p,q,''.join(cls) if n<80 else '',pick,r2,want,got)
    got2=lane_case_a2(n,cls,pick,r2,v['R'],v['M'],v['O'])
    if got2!=want:
        bad2+=1
        if bad2<6: print("MISMATCH replay",n,p,q,''.join(cls) if n<80 else '',pick,r2,want,got2)
print("total",tot,"ambiguous",amb,"bad jump",bad,"bad replay",bad2)
import sys as _sys
_sys.exit(1 if (bad or globals().get("bad2", 0)) else 0)
