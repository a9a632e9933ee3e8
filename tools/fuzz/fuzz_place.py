import sys, numpy as np, collections
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob, tmc2_amd as T
from test_host_logic import _random_patch_gof
ref=ob.Reference()
stats=collections.Counter()
FIELDS=("index","viewId","u1","v1","sizeU","sizeV","sizeU0","sizeV0","u0","v0","patchOrientation")
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng=np.random.default_rng(1000+seed)
    frames=int(rng.integers(2,7))
    gof=_random_patch_gof(rng, frames, int(rng.integers(3,40)), drift=int(rng.integers(0,30)), churn=float(rng.choice([0.0,0.1,0.4])))
    min_w=int(rng.choice([128,256,512,1280])); min_h=int(rng.choice([64,128,256,512,1280]))
    for mode in (1,2):
        try:
            got=T.host_pack_gof_records(gof,mode,min_w,min_h)          # tmc2_host_place_segments
        except T.Tmc2Error:
            stats["refused"]+=1; continue                               # undefined / never returning in the reference: not run there
        exp,rc=ref.place_records(gof,min_w,min_h,mode)                  # PCCEncoder::placeSegments itself
        canvas=T.encoder_canvas_size([max([g[4] for g in got]+([min_h] if mode==2 else []))], max([min_w]+[g[3] for g in got]), min_w, min_h)
        ok=tuple(canvas)==tuple(rc)
        for (gl,gpool,gm,_,_),(el,eo,em) in zip(got,exp):
            if len(gl)!=len(el): ok=False; continue
            for n in FIELDS:
                if n=="index" and mode==1: continue
                if not np.array_equal(gl[n],el[n]): ok=False
            if not np.array_equal(gm,em): ok=False
            mine=np.concatenate([gpool[p["occOffset"]:p["occOffset"]+p["sizeU0"]*p["sizeV0"]] for p in gl]) if len(gl) else np.zeros(0,np.uint8)
            if not np.array_equal(mine,eo[:len(mine)]): ok=False
        stats["mode%d_%s"%(mode,"ok" if ok else "MISMATCH")]+=1
        if not ok: print("MISMATCH seed",seed,"mode",mode,canvas,rc)
print(dict(stats))
