"""Python check of lane_case_a_jump (n2v_unit_core.h) on rows where the SHARED class sits exactly on the row
average (excess 0; p = 1/2, q = 2, n_other = 2 n_return): the formulas with the ties taken as the exact loop takes
them, against the pairing loop of generate_alias_tables.  Prints total / bad (must be 0)."""
import os, random
def ref_tables(w):
    n=len(w); alias=[0]*n; avg=sum(w)/n; probs=[x/avg for x in w]
    under=[i for i in range(n) if probs[i]<1.0]; over=[i for i in range(n) if not probs[i]<1.0]
    while under and over:
        u,o=under.pop(),over.pop(); alias[u]=o; probs[o]=probs[o]+probs[u]-1.0
        (under if probs[o]<1.0 else over).append(o)
    return alias,probs
def a_flat(n, cls, pick, r2, gR, gM, gO):
    lst=[i for i in range(n) if cls[i]=='M']; R=[i for i in range(n) if cls[i]=='R']
    nR,nM=len(R),len(lst); nO=n-nR-nM; rpos=R[0] if R else 0
    isum=nR*gR+nM*gM+nO*gO; EM=gM*n-isum; ER=gR*n-isum; D=isum-gO*n
    assert EM==0 and ER>0 and D>0
    mA=sum(1 for x in lst if x>rpos) if nR else nM
    N=nM+nR
    def pos_of(i):
        if i<=mA: return lst[nM-i]
        if i<=mA+nR: return rpos+nR-(i-mA)
        return lst[nM-(i-nR)]
    def X_of(i):
        if i<=mA: return 0
        if i<=mA+nR: return (i-mA)*ER
        return nR*ER
    cd=lambda a,b:-(-a//b)
    if cls[pick]=='O':
        above=sum(1 for x in range(pick+1,n) if cls[x]!='O'); r=(n-1-pick)-above; T=r*D
        if T<=0: i=1
        elif nR>0 and nR*ER>=T: i=mA+cd(T,ER)
        else: return None
        return pos_of(i)
    if cls[pick]=='R': i0=mA+(nR-(pick-rpos))
    else:
        d=nM-lst.index(pick); i0=d if d<=mA else d+nR
    if i0>N: return None
    if i0==N: return pick
    X=X_of(i0)
    if X>=nO*D: return pick        # every underfull slot is absorbed by then: never demoted
    rem=X%D
    prob=1.0+(rem-D)/isum
    return pick if r2<prob else pos_of(i0+1)
random.seed(5); bad=tot=0
for trial in range(int(os.environ.get("N2V_MODEL_TRIALS", 200000))):
    nR=random.choice([1,1,1,2,3]); nO=2*nR   # p=0.5,q=2: bR=2,bM=1,bO=0.5: avg=1 <=> nO = 2 nR
    nM=random.randint(1,40); n=nR+nM+nO
    cls=['M']*n; rp=random.randint(0,n-nR)
    for k in range(nR): cls[rp+k]='R'
    free=[i for i in range(n) if cls[i]=='M']
    for i in random.sample(free,nO): cls[i]='O'
    b={'R':2.0,'M':1.0,'O':0.5}; w=[b[c] for c in cls]
    alias,probs=ref_tables(w); avg=sum(w)/n; assert avg==1.0
    for pick in range(n):
        r2=random.getrandbits(32)/2**32
        p0=w[pick]/avg
        if p0<1.0 and r2<p0: continue
        want=pick if r2<probs[pick] else alias[pick]
        got=a_flat(n,cls,pick,r2,4,2,1)
        tot+=1
        if got!=want:
            bad+=1
            if bad<6: print("MISMATCH",''.join(cls),pick,r2,want,got,probs[pick],alias[pick])
print("total",tot,"bad",bad)
import sys; sys.exit(1 if bad else 0)
