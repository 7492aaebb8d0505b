"""Python model of lane_case_b_jump ("other" is the only overfull class) in node2vec_amd/csrc/n2v_unit_core.h, checked against the pairing loop
of generate_alias_tables (reference randomwalk.py:175-189, restated in ref_tables) on random rows of
the three class values: `python case_b_jump.py` (short rows), `python case_b_jump.py big` (long rows); N2V_MODEL_TRIALS
overrides the number of rows.  Prints the mismatch count (must be 0) and how many rows the closed
form leaves to the replay ("ambiguous")."""
import os
import random, math
def ref_tables(w):
    n=len(w); alias=[0]*n; avg=sum(w)/n; probs=[x/avg for x in w]
    under=[i for i in range(n) if probs[i]<1.0]; over=[i for i in range(n) if not probs[i]<1.0]
    while under and over:
        u,o=under.pop(),over.pop(); alias[u]=o; probs[o]=probs[o]+probs[u]-1.0
        (under if probs[o]<1.0 else over).append(o)
    return alias,probs
def jump_b(n,cls,pick,r2,gR,gM,gO):
    Rpos=[i for i in range(n) if cls[i]=='R']; lst=[i for i in range(n) if cls[i]=='M']
    nR,nM=len(Rpos),len(lst); nO=n-nR-nM; rpos=Rpos[0] if Rpos else 0
    isum=nR*gR+nM*gM+nO*gO; e=gO*n-isum; dR=isum-gR*n; dM=isum-gM*n
    if e<=0 or (nM and dM<=0) or (nR and dR<=0) or nO==0: return None
    mA=sum(1 for x in lst if x>rpos) if nR else nM
    S=nM+nR
    def Y_of(j):
        if j<=mA: return j*dM
        if j<=mA+nR: return mA*dM+(j-mA)*dR
        return mA*dM+nR*dR+(j-mA-nR)*dM
    def specials_ge(p):   # number of specials with position >= p
        return sum(1 for x in lst if x>=p)+max(0,min(nR,rpos+nR-p))
    def opos(t):          # position of the t-th highest O slot
        c=0
        for _ in range(200):
            c2=specials_ge(n-t-c)
            if c2==c: return n-t-c
            c=c2
        return None
    if cls[pick]!='O':
        if cls[pick]=='R': j=mA+(nR-(pick-rpos))
        else:
            d=nM-lst.index(pick); j=d if d<=mA else d+nR
        if j==1: t=1
        else:
            Yp=Y_of(j-1)
            t=-(-Yp//e)
            if t*e==Yp: return None
        if t<1 or t>nO: return None
        return opos(t)      # r2 >= value of the special (quick exit took the other case)
    t=(n-pick)-specials_ge(pick+1)   # rank of pick among O slots, 1-based from the top  (pick itself is O)
    if t>nO: return None
    if t==nO: return pick
    T=t*e
    # smallest j with Y_j > T
    if mA>0 and mA*dM>T: j=T//dM+1
    else:
        Y1=mA*dM
        if nR>0 and Y1+nR*dR>T: j=mA+(T-Y1)//dR+1
        else: j=mA+nR+(T-Y1-nR*dR)//dM+1
    if j<1 or j>S: return None
    if j>1 and Y_of(j-1)==T: return None
    rem=T-Y_of(j)     # < 0
    prob=1.0+rem/isum
    if abs(prob-r2)<1e-9: return None
    return pick if r2<prob else opos(t+1)
random.seed(2); bad=0; amb=0; tot=0
import sys
big = len(sys.argv)>1
for trial in range(int(os.environ.get("N2V_MODEL_TRIALS", 6000 if big else 200000))):
    n=random.choice([100,300,1000,3000]) if big else random.randint(2,60)
    p,q=random.choice([(4.0,0.25),(2.0,0.5),(1.0,0.5),(0.5,0.25),(2.0,0.25)])
    bR,bM,bO=1/p,1.0,1/q
    TR,TM,TO=int(bR*2**20),int(bM*2**20),int(bO*2**20); g=math.gcd(TR,math.gcd(TM,TO))
    cls=['O']*n
    nR=random.choice([0,1,1,1,2]); rp=random.randint(0,n-nR)
    for k in range(nR): cls[rp+k]='R'
    for i in range(n):
        if cls[i]=='O' and random.random()<random.choice([0.02,0.1,0.3]): cls[i]='M'
    w=[{'R':bR,'M':bM,'O':bO}[c] for c in cls]
    avg=sum(w)/n; v={'R':bR/avg,'M':bM/avg,'O':bO/avg}
    cnt={c:cls.count(c) for c in 'RMO'}
    if not (cnt['O']>0 and not v['O']<1 and (not cnt['R'] or v['R']<1) and (not cnt['M'] or v['M']<1) and (cnt['R'] or cnt['M'])): continue
    alias,probs=ref_tables(w)
    pick=random.randrange(n); r2=random.getrandbits(32)/2**32
    p0=v[cls[pick]]
    if p0<1.0 and r2<p0: continue
    want = pick if r2<probs[pick] else alias[pick]
    got=jump_b(n,cls,pick,r2,TR//g,TM//g,TO//g)
    tot+=1
    if got is None: amb+=1
    elif got!=want:
        bad+=1
        if bad<6: print("MISMATCH",n,p,q,''.join(cls) if n<70 else '',pick,r2,want,got)
print("total",tot,"ambiguous",amb,"bad",bad)
import sys as _sys
_sys.exit(1 if (bad or globals().get("bad2", 0)) else 0)
