import sys, numpy as np, collections, time
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob, tmc2_amd as T
oracle=ob.Oracle(); ref=ob.Reference()
stats=collections.Counter()
def bits(a): return np.ascontiguousarray(a,dtype=np.float64).view(np.uint64)
def cloud(rng):
    kind=int(rng.integers(0,6)); n=int(rng.integers(40,3000))
    if kind==0: base=rng.integers(0,60,(n,3))                         # dense cube: many duplicates removed below
    elif kind==1: base=rng.integers(0,400,(n,3)); base[:,2]=7           # a plane (degenerate covariance)
    elif kind==2: base=np.stack([np.arange(n),np.arange(n)//2,np.full(n,3)],1)   # a line
    elif kind==3: base=rng.integers(0,1000,(n,3))                       # dust
    elif kind==4:                                                       # two sheets close together
        base=rng.integers(0,120,(n,3)); base[:,1]=np.where(rng.random(n)<0.5,10,12)
    else:                                                               # lattice: exact distance ties everywhere
        g=np.stack(np.meshgrid(np.arange(12),np.arange(12),np.arange(max(1,n//144))),-1).reshape(-1,3)*int(rng.integers(1,4)); base=g
    xyz=np.unique(np.clip(base,0,2047).astype(np.int16),axis=0)
    xyz=xyz[rng.permutation(len(xyz))]
    return np.ascontiguousarray(xyz)
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    rng=np.random.default_rng(9000+seed)
    xyz=cloud(rng)
    if len(xyz)<20: continue
    ok=True; why=[]
    # tree order: product host builder vs oracle ; knn: oracle vs reference
    perm,_,_=T.host_kdtree_build(xyz)
    if not np.array_equal(perm,oracle.kdtree_perm(xyz)[0]): ok=False; why.append("tree")
    k=16 if len(xyz)>=16 else 8
    a=oracle.knn_self(xyz,k); b=ref.knn_self(xyz,k)
    if not np.array_equal(a,b): ok=False; why.append("knn")
    if k==16:
        no=oracle.normals(xyz,16,True); nr=ref.normals(xyz,16,True)
        if not np.array_equal(bits(no),bits(nr)): ok=False; why.append("normals %d"%int((bits(no)!=bits(nr)).any(1).sum()))
        hn=T.host_orient_normals(xyz,a,oracle.compute_normals(xyz,a))
        if not np.array_equal(bits(hn),bits(nr)): ok=False; why.append("host_orient %d"%int((bits(hn)!=bits(nr)).any(1).sum()))
        w=oracle.weight_normal(xyz,11,0.6); wr=ref.weight_normal(xyz,11,0.6)
        if not np.array_equal(w,wr): ok=False; why.append("weight")
        p=oracle.initial_segmentation(nr,wr); pr=ref.initial_segmentation(nr,wr)
        if not np.array_equal(p,pr): ok=False; why.append("initseg")
        q=oracle.refine_grid(xyz,nr,pr,1024,3.0,4); qr=ref.refine_grid(xyz,nr,pr,1024,3.0,4)
        if not np.array_equal(q,qr): ok=False; why.append("refine %d"%int((q!=qr).sum()))
    stats["ok" if ok else "MISMATCH"]+=1
    if not ok: print("MISMATCH",seed,len(xyz),why)
print(dict(stats))
