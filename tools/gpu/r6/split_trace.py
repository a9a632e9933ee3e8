"""Splits a rocprofv3 --kernel-trace CSV (first_pass.py --marker 1) into passes at the marker fills and prints, per pass: the
window (first start .. last end), the number of dispatches, the SUM of kernel durations, and the sum for a few kernels.
    python tools/gpu/r6/split_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys


def main():
    files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    passes, cur, started = [], [], False
    for s, e, name in rows:
        if "FillFunctor" in name or "fill" in name.lower() and "elementwise" in name:
            if started and cur:
                passes.append(cur)
            cur, started = [], True
            continue
        if started:
            cur.append((s, e, name))
    watch = ("closureKernel", "sweepKernel", "knnKernel<16, true", "pieceKernel", "lvSwapTwoKernel", "parityUnionKernel", "copyBuffer")
    print("# pass  window_ms  dispatches  sum_kernel_ms  " + "  ".join(w.split("<")[0][:14] for w in watch))
    for i, p in enumerate(passes):
        if len(p) < 1000:
            continue
        win = (max(e for _, e, _ in p) - min(s for s, _, _ in p)) / 1e6
        tot = sum(e - s for s, e, _ in p) / 1e6
        per = [sum(e - s for s, e, n in p if w in n) / 1e6 for w in watch]
        print("%5d %10.1f %11d %14.1f  " % (i, win, len(p), tot) + "  ".join("%14.2f" % x for x in per))


if __name__ == "__main__":
    main()
