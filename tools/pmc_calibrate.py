"""Known-byte-count kernels for calibrating FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md: 'calibrate on
a known byte count in your own access pattern').  Run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 28                                   # 1 GiB of int32
x = torch.arange(n, device=dev, dtype=torch.int32)
torch.cuda.synchronize()
y = x.clone()                                 # streaming: reads 1 GiB, writes 1 GiB  (elementwise copy kernel)
torch.cuda.synchronize()
m = 1 << 23
idx = (torch.arange(m, device=dev, dtype=torch.int64) * 16)        # one int32 per 64-B line
torch.cuda.synchronize()
g = x.index_select(0, idx)                    # gather: 8 Mi x 4 B useful, 8 Mi distinct 64-B lines (512 MiB of lines)
torch.cuda.synchronize()
y.index_fill_(0, idx, 7)                      # scatter: 8 Mi x 4 B writes, one per 64-B line
torch.cuda.synchronize()
print(int(g[5]), int(y[16]))
