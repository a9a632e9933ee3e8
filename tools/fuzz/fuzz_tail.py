import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob
oracle=ob.Oracle(); ref=ob.Reference()
stats=collections.Counter()
def cloud(rng):
    kind=rng.integers(0,4)
    n=int(rng.integers(50,4000))
    if kind==0:   # a thick noisy sheet
        base=rng.integers(8,200,(n,3)); base[:,2]=(base[:,0]//3+rng.integers(0,4,n))
    elif kind==1: # blobs with many duplicates
        c=rng.integers(16,300,(int(rng.integers(2,12)),3)); base=c[rng.integers(0,len(c),n)]+rng.integers(-6,7,(n,3))
    elif kind==2: # dense small cube (long candidate lists, distance ties)
        base=rng.integers(20,20+int(rng.integers(6,30)),(n,3))
    else:         # sparse
        base=rng.integers(0,1000,(n,3))
    xyz=np.clip(base,0,1023).astype(np.int16)
    bt=(rng.random(n)<rng.choice([0.1,0.5,1.0])).astype(np.uint16)
    part=rng.integers(0,int(rng.integers(1,6)),n).astype(np.uint32)
    if rng.random()<0.5:   # partitions as spatial regions (what real patches look like)
        part=((xyz[:,0]//int(rng.integers(8,64)))%5).astype(np.uint32)
    spread=int(rng.choice([5,60,30000]))
    c16=np.clip(32768+rng.integers(-spread,spread+1,(n,3)),0,65535).astype(np.uint16)
    return xyz,bt,part,c16
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    rng=np.random.default_rng(7000+seed)
    xyz,bt,part,c16=cloud(rng)
    gs=int(rng.choice([8,8,8,4,16])); thr=float(rng.choice([64,64,8,1]))
    rx,rb,rc=ref.smooth_and_transfer(xyz,bt,part,c16,gs,thr)
    ox,ob_=oracle.smooth_point_cloud_grid(xyz,bt,part,gs,thr)
    oc=oracle.transfer_colors16_bp(xyz,c16,ox,ob_)
    ok=np.array_equal(rx,ox) and np.array_equal(rb,ob_) and np.array_equal(rc,oc)
    stats["ok" if ok else "MISMATCH"]+=1
    stats["moved"]+=int((rb==3).sum())
    if not ok: print("MISMATCH",seed,len(xyz),"xyz",int((rx!=ox).any(1).sum()),"bt",int((rb!=ob_).sum()),"c16",int((rc!=oc).any(1).sum()),"moved",int((rb==3).sum()))
print(dict(stats))
