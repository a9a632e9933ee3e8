"""ASCII PLY bodies with adversarial number spellings: product reader against the reference's PCCPointSet3::read."""
import sys, numpy as np, collections, tempfile
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'mpeg-pcc-tmc2_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import oracle_binding as ob, tmc2_amd as T
ref=ob.Reference()
def spell(rng, v):
    k=int(rng.integers(0,12))
    f=float(v)+float(rng.choice([0,0,0.25,0.5,0.999,0.9999999999999999,0.000001]))
    if k==0: return "%d"%v
    if k==1: return "%.6f"%f
    if k==2: return "%.17g"%f
    if k==3: return "%e"%f
    if k==4: return "+%d"%v if v>=0 else "%d"%v
    if k==5: return "%d."%v
    if k==6: return "000%d.500"%v if v>=0 else "%d.5"%v
    if k==7: return "%dE0"%v
    if k==8: return "%.25f"%f          # > 15 significant digits: the strtod path
    if k==9: return "%dabc"%v           # atof stops at the first bad character
    if k==10: return "%.3fe+00"%f
    return float(f).hex() if rng.random()<0.5 else "%d.0e-0"%v
stats=collections.Counter()
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(int(sys.argv[1]),int(sys.argv[2])):
        rng=np.random.default_rng(21000+seed)
        n=int(rng.integers(1,400))
        xyz=rng.integers(-5,2000,(n,3)); rgb=rng.integers(0,256,(n,3))
        sep=lambda: str(rng.choice([" ","  ","\t"," \t "]))
        eol=str(rng.choice(["\n","\r\n"]))
        lines=[]
        for p,c in zip(xyz,rgb):
            cs=["%d"%c[0], "%d"%c[1] if rng.random()<0.8 else "%d.7"%c[1], "%d"%c[2] if rng.random()<0.9 else "+%d"%c[2]]
            lines.append(sep().join([spell(rng,p[0]),spell(rng,p[1]),spell(rng,p[2])]+cs)+(sep() if rng.random()<0.3 else ""))
            if rng.random()<0.02: lines.append(str(rng.choice([""," ","\t"])))
        head="ply\nformat ascii 1.0\ncomment x\nelement vertex %d\nproperty %s x\nproperty %s y\nproperty %s z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n"%(n,rng.choice(["float","double","int"]),rng.choice(["float","double"]),rng.choice(["float","float32"]))
        path=os.path.join(tmp,"f.ply")
        open(path,"w",newline="").write(head+eol.join(lines)+eol)
        a=ref.ply_read(path); b=T.ply_read(path,False,int(rng.choice([1,3,8])))
        ok=a is not None and np.array_equal(a[0],b[0]) and np.array_equal(a[1],b[1])
        stats["ok" if ok else "MISMATCH"]+=1
        if not ok:
            bad=np.flatnonzero((a[0]!=b[0]).any(1)|(a[1]!=b[1]).any(1))[:3]
            print("MISMATCH",seed,[ (lines[i] if i<len(lines) else None, a[0][i], b[0][i], a[1][i], b[1][i]) for i in bad])
print(dict(stats))
