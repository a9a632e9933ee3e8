import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob
oracle=ob.Oracle(); ref=ob.Reference()
stats=collections.Counter()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(7000+seed)
    H=2*int(rng.integers(1,80)); W=2*int(rng.integers(1,80))
    kind=int(rng.integers(0,5))
    if kind==0: rgb=rng.integers(0,256,(3,H,W),dtype=np.uint8)
    elif kind==1: rgb=rng.choice(np.array([0,255],np.uint8),(3,H,W))                      # every clamp, both ends
    elif kind==2: rgb=np.kron(rng.integers(0,256,(3,(H+7)//8,(W+7)//8),dtype=np.uint8),np.ones((8,8),np.uint8))[:, :H, :W]
    elif kind==3: rgb=np.clip(rng.normal(128,3,(3,H,W)),0,255).astype(np.uint8)         # values at rounding boundaries
    else:
        rgb=np.zeros((3,H,W),np.uint8); rgb[:, rng.integers(0,H), :]=255; rgb[:, :, rng.integers(0,W)]=rng.integers(0,256)
    rgb=np.ascontiguousarray(rgb)
    a=oracle.convert_rgb444_to_yuv420(rgb); b=ref.convert_rgb444_to_yuv420(rgb)
    ok=all(np.array_equal(x,y) for x,y in zip(a,b))
    # the way back from arbitrary planes (what a lossy codec hands over), not only from the forward result
    y,u,v=(rng.integers(0,256,p.shape,dtype=np.uint8) for p in a) if seed%2 else a
    ok = ok and np.array_equal(oracle.convert_yuv420_to_yuv444(y,u,v), ref.convert_yuv420_to_yuv444(y,u,v))
    stats["ok" if ok else "MISMATCH"]+=1
    if not ok: print("MISMATCH",seed,H,W,kind)
print(dict(stats))
