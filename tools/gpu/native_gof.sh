#!/bin/bash
# the native C++ front end (integration/tmc2_encode_gof) timing the path on the 32-frame longdress-like GOF: one device shard with
# 16 workers, and two shards on the one GPU with 8 workers each
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$(pwd)/gpurun_out/${1:-r05}_native_front_end.txt; : > $O
make -C integration > /dev/null 2>&1
D=/tmp/gof_ply; mkdir -p $D
python - <<'PY'
import sys, os
sys.path.insert(0, "mpeg-pcc-tmc2_amd")
import multiprocessing as mp
import tmc2_amd as T
from tmc2_amd.synth import synth_cloud
def w(i):
    x, c = synth_cloud("longdress_vox10", i); T.ply_write("/tmp/gof_ply/fr_%04d.ply" % i, x, c, None, ascii=False)
with mp.get_context("fork").Pool(16) as p: p.map(w, range(32))
PY
for spec in "0 16" "0,0 8" "0,0,0,0 4"; do
  set -- $spec
  echo "== --devices $1 --workers $2" >> $O
  timeout 300 ./integration/tmc2_encode_gof --in $D/fr_%04d.ply --frames 32 --iterations 50 --voxel 4 --bits 10 --precision 4 --min-width 1280 --min-height 1280 --devices $1 --workers $2 --repeat 8 --no-tail --no-files --out /tmp/gof_out >> $O 2>&1; echo "rc=$?" >> $O
done
cat $O
