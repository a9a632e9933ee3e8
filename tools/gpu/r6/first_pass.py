"""Round 6: why is the first pass of a process over the GOF twice as slow as the steady state?  

One experiment per flag, each prints one JSON line:
    (default)            set A of 16 fresh contexts: passes 0, 1, 2 ...; then set B of 16 FRESH contexts in the same process:
                         passes 0, 1, 2 -- per-context (queue / code object / TLB first touch) or per-process?
    --preheat-ms 500     a chip-wide busy loop (torch matmuls) for that long right before pass 0 of set A -- clock ramp or not?
    --warm-one 1         ONE frame through the whole path on ONE context of the set before pass 0 (every kernel of the library has
                         been launched once in the process; fifteen contexts are untouched)
    --warm-all 1         one frame through the whole path on EVERY context, one after the other (every stream has launched every
                         kernel once; nothing has run concurrently)
    --marker 1           a torch fill kernel between passes (splits a rocprofv3 kernel trace into passes: split_trace.py)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mpeg-pcc-tmc2_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="longdress")
    ap.add_argument("--sets", type=int, default=2)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--preheat-ms", type=int, default=0)
    ap.add_argument("--warm-one", type=int, default=0)
    ap.add_argument("--warm-all", type=int, default=0)
    ap.add_argument("--marker", type=int, default=0)
    ap.add_argument("--gen-procs", type=int, default=0)
    ap.add_argument("--capacity-h", type=int, default=0, help="rows the buffers hold from the start (0: the minimum canvas)")
    ap.add_argument("--resume", type=int, default=1, help="0: rounds 4-5 -- the whole pass again after 'the GOF needs a larger canvas'")
    a = ap.parse_args()
    import bench
    from tmc2_amd import configs
    case = configs.FULL_SIZE_CASES[configs.BENCH_CONFIGS[a.config]]
    clouds = bench.start_frames(case["workload"], list(range(a.frames)), a.gen_procs)()
    import numpy as np
    import torch
    import tmc2_amd as T
    from tmc2_amd import native_gof
    native_gof.load_library()
    T.load_library().tmc2_set_host_parallelism(a.workers)
    P, W0, H0 = case["precision"], case["min_w"], case["min_h"]
    packing = configs.PACKING_NAME[case["pack"]]
    out = {"config": a.config, "flags": {k: v for k, v in vars(a).items() if v}, "sets": []}
    mark = torch.zeros(1 << 20, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    keep = []
    for s in range(a.sets):
        enc = T.GofEncoder(0, a.workers, case["iterations"], case["bits3d"], P, W0, H0, timing=True, first_domain=0, vox_dim=case["vox_dim"])
        enc.reserve(max(len(c[0]) for c in clouds), W0, max(H0, W0))
        frames = enc.upload(clouds)
        cap = [W0, max(H0, a.capacity_h)]

        def bufs(W, H):
            return [(dict(occupancy=T.host_array((H, W), np.uint8), occ_video=T.host_array((H // P, W // P), np.uint8),
                          block_to_patch=T.host_array((H // 16, W // 16), np.uint32), geo0=T.host_array((H, W), np.uint16),
                          geo1=T.host_array((H, W), np.uint16)), T.host_array((2, 3, H, W), np.uint8)) for _ in frames]
        host = bufs(*cap)
        rec = {"set": s, "pass_ms": [], "stage_ms_first": None, "stage_ms_last": None}
        if s == 0 and (a.warm_one or a.warm_all):
            t0 = time.time()
            for k in range(a.workers if a.warm_all else 1):
                enc.phase_a(frames[k:k + 1], sharder=T.Sharder())
                enc.phase_b(frames[k:k + 1])
            torch.cuda.synchronize()
            rec["warm_ms"] = round(1e3 * (time.time() - t0), 1)
        if s == 0 and a.preheat_ms:
            x = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16)
            torch.cuda.synchronize()
            t0 = time.time()
            while 1e3 * (time.time() - t0) < a.preheat_ms:
                for _ in range(8):
                    x @ x
                torch.cuda.synchronize()
            rec["preheat_ms"] = round(1e3 * (time.time() - t0), 1)
        for p in range(a.passes):
            if a.marker:
                mark.fill_(s * 100 + p)
                torch.cuda.synchronize()
            enc.stage_reset()
            t0 = time.time()
            resume = False
            marks = []
            while True:
                marks.append(round(1e3 * (time.time() - t0), 1))
                try:
                    native_gof.encode(frames, [i % a.workers for i in range(len(frames))], a.workers, case["iterations"], case["vox_dim"],
                                      case["bits3d"], P, W0, H0, packing, host, cap, resume=resume)
                    break
                except native_gof.CanvasTooSmall as e:
                    cap[:] = [max(cap[0], e.size[0]), max(cap[1], e.size[1])]
                    t1 = time.time()
                    host = bufs(*cap)
                    rec["canvas_grew_in_pass"] = p
                    rec["new_buffers_ms"] = round(1e3 * (time.time() - t1), 1)
                    resume = bool(a.resume)
            marks.append(round(1e3 * (time.time() - t0), 1))
            torch.cuda.synchronize()
            rec["pass_ms"].append(round(1e3 * (time.time() - t0), 1))
            rec.setdefault("call_marks_ms", []).append(marks)
            st = {k: round(v / len(frames), 2) for k, v in enc.stage_ms().items() if not k.startswith(("refine_row", "refine_vox", "refine_sweeps_ex"))}
            if p == 0:
                rec["stage_ms_first"] = st
            rec["stage_ms_last"] = st
        rec["pool"] = enc.pool_stats()
        out["sets"].append(rec)
        keep.append((enc, frames, host))                        # (set A stays alive while set B runs: its memory is not handed over)
    if a.marker:
        mark.fill_(9999)
        torch.cuda.synchronize()
    print(json.dumps(out))
    sys.stdout.flush()
    for enc, frames, _ in keep:
        for fr in frames:
            fr.close()
        enc.close(join=True)


if __name__ == "__main__":
    main()
