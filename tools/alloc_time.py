#!/usr/bin/env python3
"""How long hipMalloc takes for buffers of the size of the per-chunk job buffers (run on the GPU box)."""
import time, torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for gb in (8, 48, 96, 160):
    t = time.time(); x = torch.empty(int(gb * 2 ** 30), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.time() - t
    t = time.time(); x[:: 1 << 21].fill_(1); torch.cuda.synchronize(); t2 = time.time() - t   # touch one byte per 2 MiB
    t = time.time(); del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t3 = time.time() - t
    print("%4d GiB: malloc %.3f s, touch %.3f s, free %.3f s" % (gb, t1, t2, t3), flush=True)
