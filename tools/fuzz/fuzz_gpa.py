import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob, tmc2_amd as T
from test_host_logic import _random_patch_gof
oracle=ob.Oracle()
stats=collections.Counter()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(1000+seed)
    frames=int(rng.integers(2,7))
    gof=_random_patch_gof(rng, frames, int(rng.integers(3,40)), drift=int(rng.integers(0,30)), churn=float(rng.choice([0.0,0.1,0.4])))
    min_w=int(rng.choice([128,256,512,1280])); min_h=int(rng.choice([64,128,256,512,1280]))
    per=[]; bad=False
    for rec,occ in gof:
        if per:
            _,pplaced,porder,_=per[-1]
            e=oracle.pack_spatial_consistency(rec,occ,pplaced[porder],min_w)
            try:
                g=T.host_pack_spatial_consistency(rec,occ,pplaced[porder],min_w)
            except T.Tmc2Error:
                g=None
            if e is None or g is None:
                stats["sc_runaway"]+=1
                if (e is None)!=(g is None): print("MISMATCH sc runaway",seed)
                bad=True; break
            placed,order,match,h=g
            ep,eo,em,eh=e
            if not (h==eh and np.array_equal(order,eo) and np.array_equal(match,em) and all(np.array_equal(placed[k],ep[k]) for k in ("u0","v0","patchOrientation"))):
                print("MISMATCH sc",seed); bad=True; break
        else:
            placed,order,h=oracle.pack_flexible(rec,occ,min_w); match=np.full(len(order),-1,np.int32)
        per.append((dict(occupancy=occ,matches=match),placed,order,h))
    if bad: continue
    exp=oracle.global_patch_allocation(per,min_w,min_h)
    tw,th=oracle.tile_size(per,min_w,min_h)
    args=([placed[order] for _,placed,order,_ in per],[seg["occupancy"] for seg,_,_,_ in per],[seg["matches"] for seg,_,_,_ in per],tw,th,min_w,min_h)
    try:
        got=T.host_global_patch_allocation(*args)
    except T.Tmc2Error as e:
        got=None
    if exp is None or got is None:
        stats["gpa_refused"]+=1
        if (exp is None)!=(got is None): print("MISMATCH gpa refusal",seed, exp is None, got is None)
        continue
    ok=True
    for f,((gl,go,gm,gw,gh),(el,eo,em,ew,eh)) in enumerate(zip(got,exp)):
        if (gw,gh)!=(ew,eh) or not np.array_equal(gm,em) or any(not np.array_equal(gl[n],el[n]) for n in el.dtype.names) or not np.array_equal(go,eo[:len(go)]): ok=False
    stats["gpa_ok" if ok else "gpa_MISMATCH"]+=1
    if not ok: print("MISMATCH gpa",seed)
    if any((gm>=0).any() for _,_,gm,_,_ in got): stats["with_matches"]+=1
print(dict(stats))
