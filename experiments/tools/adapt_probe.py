import sys, torch
sys.path.insert(0, '/root/repo')
from vod_amd.index import HipFlatIndex, PackedTopk
n, d, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, 1024, 512, 200
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == 'f16') else torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(1)
ix = HipFlatIndex(d, n, dtype=dt, device=0, exact_f32=True)
for c in range(2):
    ix.add(torch.randn((n // 2, d), generator=g, device='cuda'))
q = torch.randn((nq, d), generator=g, device='cuda')
print("sync searches:")
for it in range(6):
    s, i = ix.search(q, k)
    print(it, "kx", ix.get_stat("last_exact_kx"), "need", ix.get_stat("last_exact_need"), "band", ix.get_stat("last_exact_band_queries"), "outliers", ix.get_stat("exact_outliers"))
print("pipelined (one ahead):")
p = [PackedTopk(nq, k, torch.device('cuda', 0)) for _ in range(2)]
ix.search_async(q, k, out=(p[0].scores, p[0].ids))
for it in range(8):
    ix.search_async(q, k, out=(p[(it + 1) % 2].scores, p[(it + 1) % 2].ids))
    ix.finish()
    print(it, "kx", ix.get_stat("last_exact_kx"), "need", ix.get_stat("last_exact_need"))
ix.finish()
