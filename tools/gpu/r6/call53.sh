#!/bin/bash
# round 6, call 53: the hooks fuzzer's seed 232 (a plane of 82 points, duplicates with other colours in the source): the colour transfer with and without the split searches against the oracle
export TMPDIR=/tmp; mkdir -p gpurun_out
python - > gpurun_out/r06c53_seed232.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "mpeg-pcc-tmc2_amd"); sys.path.insert(0, "tests")
import oracle_binding as ob, tmc2_amd as T
orc = ob.Oracle(); ctx = T.Context(0)
d = np.load("/tmp/transfer_seed232.npz"); src, col, tgt = d["src"], d["col"], d["tgt"]
exp = orc.transfer_colors(src, col, tgt)
k8 = orc.knn(src, tgt, 8); b1 = orc.knn(tgt, src, 1)
for form in (None, "0"):
    ctx.set_option("KNN_SPLIT", form)
    got = ctx.transfer_colors(src, col, tgt)
    bad = np.nonzero((got != exp).any(1))[0]
    print("KNN_SPLIT", form, "differing targets:", len(bad))
    for t in bad[:12]:
        same = (src[k8[t]] == tgt[t]).all(1)
        voters = np.nonzero(b1[:, 0] == t)[0]
        print("  target", t, tgt[t], "got", got[t], "exp", exp[t], "k8", k8[t], "identical", same.astype(int), "dup targets at this position", int((tgt == tgt[t]).all(1).sum()), "voters", voters[:8])
    # the searches themselves
    g8 = ctx.frame(src).kdtree_search(tgt, 8) if hasattr(T.Frame, "kdtree_search") else None
    if g8 is not None:
        print("  knn8 rows equal the oracle's:", bool(np.array_equal(g8, k8)))
    g1 = ctx.frame(tgt).kdtree_search(src, 1)
    print("  knn1 (source in target) equal the oracle's:", bool(np.array_equal(g1, b1)), "differing", np.nonzero((g1 != b1).any(1))[0][:10])
PY
cat gpurun_out/r06c53_seed232.txt | cut -c1-300
