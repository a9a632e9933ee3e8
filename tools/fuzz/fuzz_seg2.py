import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)),'fuzz_seg.py')).read().split("for seed in range")[0].split("stats=collections.Counter()")[1])
oracle=ob.Oracle(); ref=ob.Reference()
stats=collections.Counter()
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    rng=np.random.default_rng(11000+seed)
    xyz=cloud(rng)
    if len(xyz)<64: continue
    rgb=rng.integers(0,256,(len(xyz),3),dtype=np.uint8)
    why=[]
    w=ref.weight_normal(xyz,11,0.6)
    sp=ob.seg_params(int(rng.integers(1,6)),11,w)
    a=oracle.segment(xyz,rgb,sp); b=ref.segment(xyz,rgb,sp)
    for k in b:
        x,y=a[k],b[k]
        if isinstance(y,np.ndarray) and y.dtype.names:
            for n in y.dtype.names:
                if n not in ("depthOffset","occOffset") and not np.array_equal(x[n],y[n]): why.append("seg."+n)
        elif isinstance(y,np.ndarray):
            if x.shape!=y.shape or not np.array_equal(x,y): why.append("seg."+k)
    # S18 with duplicate targets / sources
    tgt=xyz[rng.integers(0,len(xyz),len(xyz)//2)]+rng.integers(-1,2,(len(xyz)//2,3)); tgt=np.clip(tgt,0,2047).astype(np.int16)
    if not np.array_equal(oracle.transfer_colors(xyz,rgb,tgt),ref.transfer_colors(xyz,rgb,tgt)): why.append("transfer")
    # S23 with duplicates on the reconstruction side
    rc=rng.integers(0,256,(len(tgt),3),dtype=np.uint8)
    nrm=ref.normals(xyz,16,True) if len(xyz)>=16 else None
    qa,ca=oracle.metrics(xyz,rgb,tgt,rc,nrm); qb,cb=ref.metrics(xyz,rgb,tgt,rc,nrm)
    if not (np.array_equal(qa.view(np.uint64),qb.view(np.uint64)) and np.array_equal(ca,cb)): why.append("metrics")
    stats["ok" if not why else "MISMATCH"]+=1
    stats["patches"]+=len(b["patches"])
    if why: print("MISMATCH",seed,len(xyz),sorted(set(why)))
print(dict(stats))
