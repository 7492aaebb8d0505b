"""Python check of the shared-stack closed form (lane_case_b2_jump, n2v_unit_core.h) on rows where "other" sits
exactly on the row average (excess 0; p = 1/4, q = 1/2, n_shared = 2 n_return): the formulas with the ties taken as
the exact loop takes them, against the pairing loop of generate_alias_tables.  Prints total / bad (must be 0)."""
import random, math
def ref_tables(w):
    n=len(w); alias=[0]*n; avg=sum(w)/n; probs=[x/avg for x in w]
    under=[i for i in range(n) if probs[i]<1.0]; over=[i for i in range(n) if not probs[i]<1.0]
    while under and over:
        u,o=under.pop(),over.pop(); alias[u]=o; probs[o]=probs[o]+probs[u]-1.0
        (under if probs[o]<1.0 else over).append(o)
    return alias,probs
def b2_flat(n, cls, pick, r2, gR, gM, gO):
    lst=[i for i in range(n) if cls[i]=='M']; R=[i for i in range(n) if cls[i]=='R']
    nR,nM=len(R),len(lst); nO=n-nR-nM; rpos=R[0]
    isum=nR*gR+nM*gM+nO*gO; e=gO*n-isum; eR=gR*n-isum; dM=isum-gM*n
    assert e==0 and eR>0 and dM>0
    lower=lambda pos: sum(1 for x in lst if x<pos)
    mA=nM-lower(rpos); rho=(n-rpos-nR)-mA; nS=nO+nR
    def stack_pos(t):
        c=0
        for _ in range(64):
            c2=nM-lower(n-t-c)
            if c2==c: return n-t-c
            c=c2
    def Xo(t):
        if t<=rho: return 0
        if t<=rho+nR: return (t-rho)*eR
        return nR*eR
    cd=lambda a,b:-(-a//b)
    if cls[pick]=='M':
        j=nM-lower(pick)
        if j==1: t=1
        else:
            Yp=(j-1)*dM
            if nR*eR>=Yp: t=rho+cd(Yp,eR)
            else: return None
        return stack_pos(t)
    if cls[pick]=='R': t=rho+(rpos+nR-pick)
    else: t=(n-pick)-(nM-lower(pick))
    if t==nS: return pick
    T=Xo(t); j=T//dM+1
    if j>nM: return pick
    prob=1.0+(T-j*dM)/isum
    if r2<prob: return pick
    return stack_pos(t+1)
random.seed(3); bad=tot=0
for trial in range(int(__import__("os").environ.get("N2V_MODEL_TRIALS", 200000))):
    nR=random.choice([1,1,1,2,3]); nM=2*nR  # p=0.25,q=0.5: bR=4,bM=1,bO=2: avg=2 <=> 4nR+nM+2nO=2n <=> nM=2nR
    nO=random.randint(1,40); n=nR+nM+nO
    cls=['O']*n; rp=random.randint(0,n-nR)
    for k in range(nR): cls[rp+k]='R'
    free=[i for i in range(n) if cls[i]=='O']; 
    for i in random.sample(free,nM): cls[i]='M'
    b={'R':4.0,'M':1.0,'O':2.0}; w=[b[c] for c in cls]
    alias,probs=ref_tables(w)
    avg=sum(w)/n; assert avg==2.0
    for pick in range(n):
        r2=random.getrandbits(32)/2**32
        p0=w[pick]/avg
        if p0<1.0 and r2<p0: continue
        want=pick if r2<probs[pick] else alias[pick]
        got=b2_flat(n,cls,pick,r2,4,1,2)
        tot+=1
        if got!=want:
            bad+=1
            if bad<6: print("MISMATCH",''.join(cls),pick,r2,want,got,probs[pick],alias[pick])
print("total",tot,"bad",bad)
