import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import functional as F, _lib
torch.manual_seed(0)
def bench(name, fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b) for a,b in v)/len(v) for k,v in _lib.TIMING.items()}
    _lib.TIMING = None
    main = max(t.values())
    print(f"{name:34s} {main*1e3:9.1f} us  {nbytes/main/1e6:8.1f} GB/s  ({nbytes/1e6:.1f} MB)  {t}")
for B, L in ((65536, 256), (16384, 256), (65536, 128), (8192, 1024)):
    p = torch.randn(B, L, device="cuda"); y = torch.randint(0,5,(B,L),device="cuda").float().sort(dim=1,descending=True)[0].contiguous()
    perm = F.shuffle_ties_order(y, 1)
    bench(f"listnet B={B} L={L}", lambda: F.listnet_loss(p, y), B*(12*L+4))
    bench(f"listmle B={B} L={L}", lambda: F.listmle_loss(p, perm), B*(20*L+4))
    bench(f"metrics B={B} L={L}", lambda: F.metrics_at_ks(p, y, [1,3,5,10,20,50], presort=True), B*(8*L+4*6*4))
    bench(f"lambdarank B={B} L={L}", lambda: F.lambdarank_loss(p, y), B*(12*L+4))
    bench(f"shuffle_ties B={B} L={L}", lambda: F.shuffle_ties_order(y, 2), B*(12*L))
