"""Python model of lane_case_b (run-by-run replay of the mirror arrangement) in node2vec_amd/csrc/n2v_unit_core.h, checked against the pairing loop
of generate_alias_tables (reference randomwalk.py:175-189, restated in ref_tables) on random rows of
the three class values: `python case_b_replay.py` (short rows), `python case_b_replay.py big` (long rows); N2V_MODEL_TRIALS
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
def lane_case_b(n,cls,pick,r2,vR,vM,vO):
    Rpos=[i for i in range(n) if cls[i]=='R']; lst=[i for i in range(n) if cls[i]=='M']
    nR,nM=len(Rpos),len(lst); nO=n-nR-nM; rpos=Rpos[0] if Rpos else 0
    pickR=cls[pick]=='R'; pickM=cls[pick]=='M'
    def specials_ge(p): return sum(1 for x in lst if x>=p)+max(0,min(nR,rpos+nR-p))
    def other_pos(t):
        if t<1 or t>nO: return 0
        c=0
        for _ in range(64):
            c2=specials_ge(n-t-c)
            if c2==c: return n-t-c
            c=c2
        raise RuntimeError
    pick_rank=0
    if not pickR and not pickM: pick_rank=(n-pick)-specials_ge(pick+1)
    d=vO-1.0; inv=1.0/d if d>0 else 0.0
    km=nM-1; kr=nR-1; t_used=0; have_cur=False; cur_val=0.0
    while True:
        pm=lst[km] if km>=0 else -1; pr=rpos+kr if kr>=0 else -1
        if pm<0 and pr<0: break
        if not have_cur and t_used>=nO: break
        if pm>pr: ui,uv=pm,vM; km-=1
        else: ui,uv=pr,vR; kr-=1
        over_rank=t_used+1
        if ui==pick: return other_pos(over_rank)
        a=(cur_val if have_cur else vO)+uv-1.0
        if not (a<1.0): cur_val=a; have_cur=True; continue
        if pick_rank==over_rank: return pick if r2<a else other_pos(over_rank+1)
        t_used=over_rank; have_cur=False
        avail=nO-t_used
        if avail<=0: break
        a1=vO+a-1.0
        if not (a1<1.0): cur_val=a1; have_cur=True; continue
        need=1.0-a1; m1=float(avail-1)
        if m1*d<need:
            if pick_rank>t_used:
                pv=a1+float(pick_rank-t_used-1)*d
                return pick if r2<pv else other_pos(pick_rank+1)
            break
        j=min(max(math.ceil(need*inv),1.0),m1)
        while j*d<need: j+=1.0
        while j>=2.0 and (j-1.0)*d>=need: j-=1.0
        jd=int(j)
        if pick_rank>t_used and pick_rank<=t_used+jd:
            pv=a1+float(pick_rank-t_used-1)*d
            return pick if r2<pv else other_pos(pick_rank+1)
        a_prev=a1+(j-1.0)*d
        cur_val=vO+a_prev-1.0
        t_used+=jd; have_cur=True
    return 0 if (pickR or pickM) else pick
import sys
random.seed(5); bad=0; tot=0
big=len(sys.argv)>1
for trial in range(int(os.environ.get("N2V_MODEL_TRIALS", 4000 if big else 150000))):
    n=random.choice([100,400,1500,4000]) if big else random.randint(2,80)
    p,q=random.choice([(4.0,0.25),(2.0,0.5),(1.0,0.5),(0.5,0.25),(2.0,0.25)])
    bR,bM,bO=1/p,1.0,1/q
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
    got=lane_case_b(n,cls,pick,r2,v['R'],v['M'],v['O'])
    tot+=1
    if got!=want:
        bad+=1
        if bad<6: print("MISMATCH",n,p,q,''.join(cls) if n<90 else '',pick,r2,want,got,probs[pick],alias[pick])
print("total",tot,"bad",bad)
import sys as _sys
_sys.exit(1 if (bad or globals().get("bad2", 0)) else 0)
