"""How much of the chip a frame's kernels occupy, from a rocprofv3 (rocpd SQLite) kernel trace of ONE frame in flight:
per kernel, waves launched (grid / 64) against the wave slots of the chip, times its duration -- "chip-milliseconds".
A path whose launches are mostly small is latency-bound however many frames are in flight; one whose chip-milliseconds per
frame approach its wall time per frame at full throughput is occupancy-bound.
usage: python profiles/occupancy_rocpd.py <results.db> <frames profiled> [wave slots of the chip = 256 CUs x 32]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
slots = float(sys.argv[3]) if len(sys.argv) > 3 else 256 * 32
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns of the kernels view:", ", ".join(cols))
def pick(*names):
    for n in names:
        if n in cols:
            return n
    return None
gx, gy, gz = pick("grid_size_x", "grid_x"), pick("grid_size_y", "grid_y"), pick("grid_size_z", "grid_z")
if gx is None:
    print("# no grid size columns: nothing to do")
    sys.exit(0)
q = "select name, duration, %s * coalesce(%s, 1) * coalesce(%s, 1) from kernels" % (gx, gy or "1", gz or "1")
tot_t = tot_c = 0.0
by = {}
for name, dur, threads in db.execute(q):
    waves = max(1.0, threads / 64.0)
    share = min(1.0, waves / slots)
    tot_t += dur
    tot_c += dur * share
    short = re.sub(r"\(.*", "", name.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))
    a = by.setdefault(short, [0.0, 0.0, 0])
    a[0] += dur; a[1] += dur * share; a[2] += 1
print("# per frame: kernel time %.2f ms, chip-milliseconds %.2f ms (kernel time weighted by min(1, waves / %d wave slots))"
      % (tot_t / 1e6 / frames, tot_c / 1e6 / frames, slots))
print("%-46s %8s %12s %12s" % ("kernel", "calls", "time_ms", "chip_ms"))
for k, (t, c, n) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-46s %8d %12.3f %12.3f" % (k[:46], n, t / 1e6 / frames, c / 1e6 / frames))
