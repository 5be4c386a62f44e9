"""Device-to-host copy rate into page-locked memory, one copy at a time and split over two streams (tools/probes)."""
import time
import torch

for mb in (8, 56, 256):
    n = mb * 1024 * 1024 // 8
    d = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    h = torch.empty(n, dtype=torch.float64).pin_memory()
    for split in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(split)]
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            step = n // split
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    h[k * step:(k + 1) * step].copy_(d[k * step:(k + 1) * step], non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%4d MB, %d stream(s): %.2f ms = %.1f GB/s" % (mb, split, best * 1e3, mb / 1024 / best))
