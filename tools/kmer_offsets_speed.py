import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from covest_amd import kmer_hist as kh
dev = torch.device("cuda", 0)
k, L, n = 21, 100, 10_000_000
gen = torch.Generator(device=dev); gen.manual_seed(3)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
g_len = n * L // 40
genome = lut[torch.randint(0, 4, (g_len,), device=dev, generator=gen)]
starts = torch.randint(0, g_len - L, (n,), device=dev, generator=gen)
reads = genome[starts[:, None] + torch.arange(L, device=dev)[None, :]].reshape(-1).contiguous()
offs = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L)
# ragged: the same bytes cut into reads of 50 .. 150 bases
lens = torch.randint(50, 151, (int(n * L / 100 * 1.05),), device=dev, generator=gen, dtype=torch.int64)
ends = torch.cumsum(lens, 0)
n_ragged = int((ends <= n * L).sum().item())
offs_ragged = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), ends[:n_ragged]])
c = kh.KmerCounts(k, canonical=True, min_slots=1 << 20)
for layout in ("fixed", "offsets", "ragged"):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        if layout == "fixed":
            path = c.count_reads_device(reads.data_ptr(), n, L)
        elif layout == "offsets":
            path = c.count_reads_device(reads.data_ptr(), n, 0, d_offsets_ptr=offs.data_ptr(), n_bases=n * L)
        else:
            path = c.count_reads_device(reads.data_ptr(), n_ragged, 0, d_offsets_ptr=offs_ragged.data_ptr(),
                                        n_bases=int(offs_ragged[-1].item()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(layout, path, "%.1f ms" % (1e3 * dt), c.partition_info()["ms"] if path == "partitioned" else "")
