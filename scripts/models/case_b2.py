"""Python model of lane_case_b2_jump and lane_case_b2 (return run + "other" overfull, listed slots underfull) in node2vec_amd/csrc/n2v_unit_core.h, checked against the pairing loop
of generate_alias_tables (reference randomwalk.py:175-189, restated in ref_tables) on random rows of
the three class values: `python case_b2.py` (short rows), `python case_b2.py big` (long rows); N2V_MODEL_TRIALS
overrides the number of rows.  Prints the mismatch count (must be 0) and how many rows the closed
form leaves to the replay ("ambiguous")."""
import os
import random, math, sys
def ref_tables(w):
    n=len(w); alias=[0]*n; avg=sum(w)/n; probs=[x/avg for x in w]
    under=[i for i in range(n) if probs[i]<1.0]; over=[i for i in range(n) if not probs[i]<1.0]
    while under and over:
        u,o=under.pop(),over.pop(); alias[u]=o; probs[o]=probs[o]+probs[u]-1.0
        (under if probs[o]<1.0 else over).append(o)
    return alias,probs

class Geo:
    def __init__(s,n,cls):
        s.n=n; s.Rpos=[i for i in range(n) if cls[i]=='R']; s.lst=[i for i in range(n) if cls[i]=='M']
        s.nR,s.nM=len(s.Rpos),len(s.lst); s.nO=n-s.nR-s.nM; s.rpos=s.Rpos[0] if s.Rpos else 0
        lo_r=sum(1 for x in s.lst if x<s.rpos); s.mA=s.nM-lo_r
        s.rho=(n-s.rpos-s.nR)-s.mA      # "other" slots above the return run
        s.nV=s.nO+s.nR                  # overfull slots
    def specials_ge(s,p): return sum(1 for x in s.lst if x>=p)+max(0,min(s.nR,s.rpos+s.nR-p))
    def other_pos(s,t):
        if t<1 or t>s.nO: return 0
        c=0
        for _ in range(64):
            c2=s.specials_ge(s.n-t-c)
            if c2==c: return s.n-t-c
            c=c2
        raise RuntimeError
    def over_pos(s,t):      # position of the t-th overfull slot from the top (0 if none)
        if t<1 or t>s.nV: return 0
        if t<=s.rho: return s.other_pos(t)
        if t<=s.rho+s.nR: return s.rpos+s.nR-(t-s.rho)
        return s.other_pos(t-s.nR)
    def over_rank(s,pick,cls):
        if cls[pick]=='R': return s.rho+(s.rpos+s.nR-pick)
        tO=(s.n-pick)-s.specials_ge(pick+1)
        return tO if tO<=s.rho else tO+s.nR

def jump_b2(n,cls,pick,r2,gR,gM,gO):
    G=Geo(n,cls); nR,nM,nO,rho=G.nR,G.nM,G.nO,G.rho
    isum=nR*gR+nM*gM+nO*gO; e=gO*n-isum; eR=gR*n-isum; dM=isum-gM*n
    if not (nR>0 and eR>0 and e>0 and nM>0 and dM>0): return None
    def Xo(t):
        if t<=rho: return t*e
        if t<=rho+nR: return rho*e+(t-rho)*eR
        return rho*e+nR*eR+(t-rho-nR)*e
    if cls[pick]=='M':
        lo_pick=sum(1 for x in G.lst if x<pick)
        j=nM-lo_pick
        if j==1: t=1
        else:
            Yp=(j-1)*dM   # smallest t with Xo(t) >= Yp
            if rho*e>=Yp: t=-(-Yp//e)
            else:
                X1=rho*e
                if X1+nR*eR>=Yp: t=rho+(-(-(Yp-X1)//eR))
                else: t=rho+nR+(-(-(Yp-X1-nR*eR)//e))
            if Xo(t)==Yp: return None
        if t<1 or t>G.nV: return None
        return G.over_pos(t)
    t=G.over_rank(pick,cls)
    if t<1 or t>G.nV: return None
    if t==G.nV: return pick
    T=Xo(t)
    j=T//dM+1
    if j<1 or j>nM: return None
    if j>1 and (j-1)*dM==T: return None
    prob=1.0+(T-j*dM)/isum
    if abs(prob-r2)<1e-9: return None
    return pick if r2<prob else G.over_pos(t+1)

def lane_case_b2(n,cls,pick,r2,vR,vM,vO):
    G=Geo(n,cls); nR,nM,nO,rho,nV=G.nR,G.nM,G.nO,G.rho,G.nV
    pickM=cls[pick]=='M'
    pick_rank=0 if pickM else G.over_rank(pick,cls)
    d=vO-1.0; inv=1.0/d if d>0 else 0.0
    km=nM-1; t_used=0; have_cur=False; cur_val=0.0
    def in_r(t): return t>rho and t<=rho+nR
    while True:
        if km<0: break
        if not have_cur and t_used>=nV: break
        ui=G.lst[km]; uv=vM; km-=1
        over_rank=t_used+1
        if ui==pick: return G.over_pos(over_rank)
        a=(cur_val if have_cur else (vR if in_r(over_rank) else vO))+uv-1.0
        if not (a<1.0): cur_val=a; have_cur=True; continue
        if pick_rank==over_rank: return pick if r2<a else G.over_pos(over_rank+1)
        t_used=over_rank; have_cur=False
        # the rest of the demoted slot cascades down the overfull slots
        while True:
            if t_used>=nV: break
            r=t_used+1
            if in_r(r):
                val=vR+a-1.0
                if not (val<1.0): cur_val=val; have_cur=True; break
                if pick_rank==r: return pick if r2<val else G.over_pos(r+1)
                t_used=r; a=val; continue
            avail=(rho if t_used<rho else nV)-t_used
            a1=vO+a-1.0
            if not (a1<1.0): cur_val=a1; have_cur=True; break
            need=1.0-a1; m1=float(avail-1)
            if m1*d<need:
                if pick_rank>t_used and pick_rank<=t_used+avail:
                    pv=a1+float(pick_rank-t_used-1)*d
                    return pick if r2<pv else G.over_pos(pick_rank+1)
                a=a1+m1*d; t_used+=avail; continue
            j=min(max(math.ceil(need*inv),1.0),m1)
            while j*d<need: j+=1.0
            while j>=2.0 and (j-1.0)*d>=need: j-=1.0
            jd=int(j)
            if pick_rank>t_used and pick_rank<=t_used+jd:
                pv=a1+float(pick_rank-t_used-1)*d
                return pick if r2<pv else G.over_pos(pick_rank+1)
            a_prev=a1+(j-1.0)*d
            cur_val=vO+a_prev-1.0
            t_used+=jd; have_cur=True; break
        if not have_cur and t_used>=nV: break
    return 0 if pickM else pick

random.seed(11); bad=0; amb=0; tot=0; bad2=0
big=len(sys.argv)>1
for trial in range(int(os.environ.get("N2V_MODEL_TRIALS", 5000 if big else 200000))):
    n=random.choice([100,300,1000,3000]) if big else random.randint(2,70)
    p,q=random.choice([(0.25,0.5),(0.125,0.5),(0.125,0.25),(0.0625,0.5),(0.25,0.5)])
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
    if not (cnt['O']>0 and not v['O']<1 and cnt['R'] and not v['R']<1 and cnt['M'] and v['M']<1): continue
    alias,probs=ref_tables(w)
    pick=random.randrange(n); r2=random.getrandbits(32)/2**32
    p0=v[cls[pick]]
    if p0<1.0 and r2<p0: continue
    want = pick if r2<probs[pick] else alias[pick]
    got=jump_b2(n,cls,pick,r2,TR//g,TM//g,TO//g)
    tot+=1
    if got is None: amb+=1
    elif got!=want:
        bad+=1
        if bad<6: print("MISMATCH jump",n,p,q,''.join(cls) if n<80 else '',pick,r2,want,got)
    got2=lane_case_b2(n,cls,pick,r2,v['R'],v['M'],v['O'])
    if got2!=want:
        bad2+=1
        if bad2<6: print("MISMATCH replay",n,p,q,''.join(cls) if n<80 else '',pick,r2,want,got2,probs[pick],alias[pick])
print("total",tot,"ambiguous",amb,"bad jump",bad,"bad replay",bad2)
import sys as _sys
_sys.exit(1 if (bad or globals().get("bad2", 0)) else 0)
