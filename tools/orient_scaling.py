"""Host-side scaling probe: T concurrent exact orientations (S3) on ordinary (malloc) memory.
usage: python tools/orient_scaling.py [threads ...]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpeg-pcc-tmc2_amd"))
import tmc2_amd as T
from tmc2_amd.synth import synth_cloud

xyz, rgb = synth_cloud("longdress_vox10")
ctx = T.Context()
fr = ctx.frame(xyz, rgb)
fr.normals_compute_normals(16)
knn, nrm = fr.get_adjacency(16), fr.get_normals()
del fr, ctx
for threads in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64]:
    times = [0.0] * threads
    data = [(xyz.copy(), knn.copy(), nrm.copy()) for _ in range(threads)]
    def work(i):
        t = time.time()
        T.host_orient_normals(*data[i])
        times[i] = time.time() - t
    t0 = time.time()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in ths]; [t.join() for t in ths]
    wall = time.time() - t0
    print("threads %3d  per-call avg %.0f ms  max %.0f ms  wall %.0f ms  -> %.1f frames/s" %
          (threads, 1e3 * sum(times) / threads, 1e3 * max(times), 1e3 * wall, threads / wall), flush=True)
sys.stdout.flush()
os._exit(0)
