"""D2H rate of one stream into page-locked host memory, per copy size (what the canvases' way out of the timed region costs)."""
import os, sys, time, torch
print("HSA_ENABLE_SDMA", os.environ.get("HSA_ENABLE_SDMA"), "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for mb in (1, 4, 10, 23, 64):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    for _ in range(3):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    t = time.time()
    reps = 20
    for _ in range(reps):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    print("%3d MiB: %.3f ms  %.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
