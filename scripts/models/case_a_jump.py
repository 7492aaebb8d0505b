"""Python model of lane_case_a_jump ("other" is the only underfull class) in node2vec_amd/csrc/n2v_unit_core.h, checked against the pairing loop
of generate_alias_tables (reference randomwalk.py:175-189, restated in ref_tables) on random rows of
the three class values: `python case_a_jump.py` (short rows), `python case_a_jump.py big` (long rows); N2V_MODEL_TRIALS
overrides the number of rows.  Prints the mismatch count (must be 0) and how many rows the closed
form leaves to the replay ("ambiguous")."""
import os
import random, math
from fractions import Fraction
def ref_tables(w):
    n=len(w); alias=[0]*n; avg=sum(w)/n; probs=[x/avg for x in w]
    under=[i for i in range(n) if probs[i]<1.0]; over=[i for i in range(n) if not probs[i]<1.0]
    while under and over:
        u,o=under.pop(),over.pop(); alias[u]=o; probs[o]=probs[o]+probs[u]-1.0
        (under if probs[o]<1.0 else over).append(o)
    return alias,probs
def jump(n,cls,pick,r2,gR,gM,gO):
    # cls[i] in 'R','M','O'; returns result index or None (ambiguous)
    Rpos=[i for i in range(n) if cls[i]=='R']; lst=[i for i in range(n) if cls[i]=='M']
    nR,nM=len(Rpos),len(lst); nO=n-nR-nM; rpos=Rpos[0] if Rpos else 0
    isum=nR*gR+nM*gM+nO*gO; EM=gM*n-isum; ER=gR*n-isum; D=isum-gO*n
    if D<=0 or (nM and EM<=0) or (nR and ER<=0): return None
    mA = sum(1 for x in lst if x>rpos) if nR else nM
    N=nM+nR
    def pos_of(i):
        if i<=mA: return lst[nM-i]
        if i<=mA+nR: return rpos+nR-(i-mA)
        return lst[nM-(i-nR)]
    def X_of(i):
        if i<=mA: return i*EM
        if i<=mA+nR: return mA*EM+(i-mA)*ER
        return mA*EM+nR*ER+(i-mA-nR)*EM
    cd=lambda a,b:-(-a//b)
    if cls[pick]=='O':
        above=sum(1 for x in range(pick+1,n) if cls[x]!='O'); r=(n-1-pick)-above; T=r*D
        if T<=0: i=1
        elif mA>0 and mA*EM>=T: i=cd(T,EM)
        else:
            X1=mA*EM
            if nR>0 and X1+nR*ER>=T: i=mA+cd(T-X1,ER)
            else: i=mA+nR+cd(T-X1-nR*ER,EM)
        if X_of(i)==T: return None
        return pos_of(i)
    else:
        if cls[pick]=='R': j=pick-rpos; i0=mA+(nR-j)
        else:
            lo=lst.index(pick); d=nM-lo
            i0=d if d<=mA else d+nR
        if i0>N: return None
        if i0==N: return pick
        X=X_of(i0)
        if X%D==0: return None
        Kc=X//D+1; rem=X-Kc*D
        prob=1.0+rem/isum
        if abs(prob-r2)<1e-9: return None
        return pick if r2<prob else pos_of(i0+1)
random.seed(1); bad=0; amb=0; tot=0
for trial in range(int(os.environ.get("N2V_MODEL_TRIALS", 200000))):
    n=random.randint(2,60); p,q=random.choice([(0.5,2.0),(0.25,4.0),(1.0,2.0),(2.0,4.0),(0.5,1.0)])
    bR,bM,bO=1/p,1.0,1/q
    TR,TM,TO=int(bR*2**20),int(bM*2**20),int(bO*2**20); g=math.gcd(TR,math.gcd(TM,TO))
    cls=['O']*n
    nR=random.choice([0,1,1,1,2]); rp=random.randint(0,n-nR)
    for k in range(nR): cls[rp+k]='R'
    for i in range(n):
        if cls[i]=='O' and random.random()<random.choice([0.05,0.2,0.5]): cls[i]='M'
    if q==1.0: cls=[c if c!='M' else 'O' for c in cls]
    w=[{'R':bR,'M':bM,'O':bO}[c] for c in cls]
    avg=sum(w)/n; v={'R':bR/avg,'M':bM/avg,'O':bO/avg}
    cnt={c:cls.count(c) for c in 'RMO'}
    uO=v['O']<1; 
    if not(uO and cnt['O']>0 and not (cnt['R'] and v['R']<1) and not (cnt['M'] and v['M']<1) and (cnt['R'] or cnt['M'])): continue
    alias,probs=ref_tables(w)
    pick=random.randrange(n); r2=random.getrandbits(32)/2**32
    p0=v[cls[pick]]
    if p0<1.0 and r2<p0: continue
    want = pick if r2<probs[pick] else alias[pick]
    got=jump(n,cls,pick,r2,TR//g,TM//g,TO//g)
    tot+=1
    if got is None: amb+=1
    elif got!=want: bad+=1; print("MISMATCH",n,p,q,''.join(cls),pick,r2,want,got) if bad<5 else None
print("total",tot,"ambiguous",amb,"bad",bad)
import sys as _sys
_sys.exit(1 if (bad or globals().get("bad2", 0)) else 0)
