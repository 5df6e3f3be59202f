#!/usr/bin/env python3
"""Step-level model of one wave of the trace kernel: 64 lanes, each with a ray that alternates between the three traversal states
(I inner node, L leaf triangle, O object boundary); per scheduling iteration the wave runs ONE state's step for the lanes in it.
Ray shapes follow the C3 counters (9.0 inner + 1.0 mesh-leaf + 2 ground-quad leaf steps + 1 object step per ray on average; 55 % of
the rays leave the mesh's tree within 1-4 nodes).  The model reproduces the measured occupancy of the current policy (model: inner
32.8 / leaf 31.7 lanes per step; COUNT kernels: 34.6 / 34.7) and says what other policies would give:
every vote policy and repeat threshold lands within +-3 % of the current one -- the half-empty steps come from the three states,
not from the scheduling -- and a second ray per lane would fill 39.5 lanes per inner step (-16 % cycles per ray if it were free).

    python scripts/vote_policy_sim.py        (CPU only, ~1 minute)
"""
import random
def make_ray():
    seq=[]
    if random.random()<0.55:
        seq+=['I']*random.randint(1,4)
    else:
        n_in=random.randint(8,26); n_leaf=random.randint(1,4)
        pos=sorted(random.sample(range(3,n_in+n_leaf), n_leaf))
        k=0
        for i in range(n_in+n_leaf):
            if k<n_leaf and i==pos[k]: seq.append('L'); k+=1
            else: seq.append('I')
    seq+=['L','L','O']
    return seq
COST={'I':224,'L':200,'O':420}
VOTE=30
def simulate(policy, n_rays_total=150000, inner_repeat=20, leaf_repeat=4, obj_repeat=1, thresh=0):
    random.seed(1)
    lanes=[None]*64; issued=0; cycles=0; steps={'I':0,'L':0,'O':0}; lane_steps={'I':0,'L':0,'O':0}
    wait=[0]*64
    def refill():
        nonlocal issued
        for i in range(64):
            if lanes[i] is None and issued<n_rays_total:
                lanes[i]=[make_ray(),0]; issued+=1; wait[i]=0
    refill()
    def counts():
        c={'I':0,'L':0,'O':0}
        for r in lanes:
            if r is not None: c[r[0][r[1]]]+=1
        return c
    while True:
        c=counts()
        if sum(c.values())==0:
            if issued>=n_rays_total: break
            refill(); continue
        if policy=='max': s=max(c,key=lambda k:c[k])
        elif policy=='per_cost': s=max(c,key=lambda k:c[k]/COST[k])
        elif policy=='oldest':
            # state of the lane that has waited longest
            w={'I':0,'L':0,'O':0}
            for i,r in enumerate(lanes):
                if r is not None: w[r[0][r[1]]]+=wait[i]
            s=max(c,key=lambda k:(w[k] if c[k] else -1))
        elif policy=='thresh':
            # run inner while >= thresh lanes; else the fullest of the others; else inner
            if c['I']>=thresh: s='I'
            else:
                s=max(c,key=lambda k:c[k])
        rep={'I':inner_repeat,'L':leaf_repeat,'O':obj_repeat}[s]
        cycles+=VOTE
        while True:
            n=0
            for i,r in enumerate(lanes):
                if r is None: continue
                if r[0][r[1]]==s:
                    r[1]+=1; n+=1; wait[i]=0
                    if r[1]>=len(r[0]): lanes[i]=None
                else: wait[i]+=1
            steps[s]+=1; lane_steps[s]+=n; cycles+=COST[s]+10
            if s=='O': refill()
            c=counts()
            if c[s]<rep or c[s]==0: break
        if sum(1 for r in lanes if r is None)>=16: refill()
    return cycles/n_rays_total, steps, lane_steps
def show(name,res):
    cyc,st,ls=res
    print(f"{name:34s} cycles/ray {cyc:6.1f} inner {st['I']:6d} @ {ls['I']/max(1,st['I']):4.1f}  leaf {st['L']:6d} @ {ls['L']/max(1,st['L']):4.1f}  obj {st['O']:5d} @ {ls['O']/max(1,st['O']):4.1f}")
show('max (current)', simulate('max'))
show('per_cost', simulate('per_cost'))
show('oldest', simulate('oldest'))
for t in (16,24,32,40):
    show(f'inner first if >= {t}', simulate('thresh',thresh=t))
for ir,lr in ((12,4),(28,4),(20,1),(20,12),(32,16),(40,24)):
    show(f'max, inner_repeat {ir} leaf_repeat {lr}', simulate('max',inner_repeat=ir,leaf_repeat=lr))
for orr in (8,16,24):
    show(f'max, obj_repeat {orr}', simulate('max',obj_repeat=orr))
