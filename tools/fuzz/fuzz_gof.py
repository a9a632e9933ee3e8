"""Whole path S0-S22 + colour conversion + tail: oracle against the reference on GOFs of degenerate clouds."""
import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob
from test_oracle_golden import degenerate_cloud
oracle=ob.Oracle(); ref=ob.Reference()
stats=collections.Counter()
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    print("seed",seed,flush=True) if os.environ.get("FUZZ_VERBOSE") else None
    rng=np.random.default_rng(13000+seed)
    frames=[]
    for f in range(int(rng.integers(1,4))):
        xyz=degenerate_cloud(rng)
        if len(xyz)<64: break
        frames.append((xyz, rng.integers(0,256,(len(xyz),3),dtype=np.uint8)))
    if not frames: continue
    prec=int(rng.choice([4,2,1])); it=int(rng.integers(1,5))
    mode=int(rng.choice([0,1,2])) if len(frames)>1 else 0
    why=[]
    try:
        oa=oracle.phase_a(frames,it,11,prec,constrained_pack=mode)
    except Exception as e:
        stats["oracle_refused"]+=1; continue
    if oa is None: stats["oracle_refused"]+=1; continue
    ra=ref.phase_a(frames,it,11,prec,constrained_pack=mode)
    for i,(x,y) in enumerate(zip(ra,oa)):
        if (x["width"],x["height"])!=(y["width"],y["height"]): why.append("canvas")
        for k in ("occupancy","occ_video","block_to_patch","geo0","geo1"):
            if x[k].shape!=y[k].shape or not np.array_equal(x[k],y[k]): why.append("a."+k)
    if not why:
        rb=ref.phase_b(frames,ra,prec); ob_=oracle.phase_b(frames,oa,prec)
        for x,y in zip(rb,ob_):
            for k in x:
                if x[k].shape!=y[k].shape or not np.array_equal(x[k],y[k]): why.append("b."+k)
        if not why:
            dec=[np.stack([oracle.convert_yuv420_to_yuv444(*oracle.convert_rgb444_to_yuv420(b["attribute"][m])) for m in range(2)]) for b in rb]
            rc=ref.phase_c(rb,dec); oc=oracle.phase_c(oa,ob_,dec,prec)
            for x,y in zip(rc,oc):
                for k in x:
                    if not np.array_equal(x[k],y[k]): why.append("c."+k)
    stats["ok" if not why else "MISMATCH"]+=1
    if why: print("MISMATCH",seed,"frames",[len(f[0]) for f in frames],"prec",prec,"mode",mode,sorted(set(why)))
print(dict(stats))
