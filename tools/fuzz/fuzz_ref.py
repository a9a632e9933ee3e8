import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob
from test_host_logic import _random_patch_gof
oracle=ob.Oracle(); ref=ob.Reference()
stats=collections.Counter()
FIELDS=("index","viewId","u1","v1","sizeU","sizeV","sizeU0","sizeV0","u0","v0","patchOrientation")
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(1000+seed)
    frames=int(rng.integers(2,7))
    gof=_random_patch_gof(rng, frames, int(rng.integers(3,40)), drift=int(rng.integers(0,30)), churn=float(rng.choice([0.0,0.1,0.4])))
    min_w=int(rng.choice([128,256,512,1280])); min_h=int(rng.choice([64,128,256,512,1280]))
    for mode in (1,2):
        per=[]; bad=False
        for rec,occ in gof:
            if per:
                _,pplaced,porder,_=per[-1]
                e=oracle.pack_spatial_consistency(rec,occ,pplaced[porder],min_w)
                if e is None: bad=True; break
                placed,order,match,h=e
            else:
                placed,order,h=oracle.pack_flexible(rec,occ,min_w); match=np.full(len(order),-1,np.int32)
            per.append((dict(occupancy=occ,matches=match),placed,order,h))
        if bad: stats["skipped_runaway"]+=1; continue
        if mode==2:
            exp=oracle.global_patch_allocation(per,min_w,min_h)
            if exp is None: stats["skipped_undefined"]+=1; continue
            lists=[(l,o,m) for l,o,m,_,_ in exp]
            tw,th=max([g[3] for g in exp]+[min_w]),max([g[4] for g in exp]+[min_h])
            canvas=oracle.gof_canvas_size([th],tw,min_w,min_h)
        else:
            lists=[]
            for seg,placed,order,h in per:
                l=placed[order]; pool=np.concatenate([seg["occupancy"][p["occOffset"]:p["occOffset"]+p["sizeU0"]*p["sizeV0"]] for p in l]) if len(l) else np.zeros(1,np.uint8)
                lists.append((l,pool,seg["matches"]))
            canvas=oracle.gof_canvas_size([x[3] for x in per],oracle.tile_size(per,min_w,min_h)[0],min_w,min_h)
        got,rc=ref.place_records(gof,min_w,min_h,mode)
        ok = tuple(canvas)==tuple(rc)
        for (el,eo,em),(gl,go,gm) in zip(lists,got):
            if len(el)!=len(gl): ok=False; continue
            for n in FIELDS:
                if n=="index" and mode==1: continue
                if not np.array_equal(el[n],gl[n]): ok=False
            if not np.array_equal(em,gm): ok=False
            k=int((el["sizeU0"]*el["sizeV0"]).sum())
            if not np.array_equal(eo[:k],go[:k]): ok=False
        stats["mode%d_%s"%(mode,"ok" if ok else "MISMATCH")]+=1
        if not ok:
            det=[]
            for f,((el,eo,em),(gl,go,gm)) in enumerate(zip(lists,got)):
                if len(el)!=len(gl): det.append((f,"count")); continue
                d=[n for n in FIELDS if not (n=="index" and mode==1) and not np.array_equal(el[n],gl[n])]
                if not np.array_equal(em,gm): d.append("match")
                if d: det.append((f,d))
            print("MISMATCH seed",seed,"mode",mode, canvas, rc, det, "oracle tile heights", [g[4] for g in exp] if mode==2 else None)
print(dict(stats))
