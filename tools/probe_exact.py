import numpy as np, torch, time, sys
sys.path.insert(0,'/root/repo')
from vod_amd.index import HipFlatIndex
from oracle.flat_ip import flat_ip_topk
def run(N,D,nq,k,dt,seed=0, expand=None, cap=None):
    g=torch.Generator().manual_seed(seed)
    x=torch.randn(N,D,generator=g); q=torch.randn(nq,D,generator=g)
    with HipFlatIndex(D,N,dtype=dt,device=0,exact_f32=True) as ix:
        ix.add(x.numpy())
        if expand: ix.set_param("exact_expand",expand)
        if cap: ix.set_param("cand_cap",cap)
        qd=q.cuda()
        s,i=ix.search(qd,k)
        torch.cuda.synchronize(); t0=time.time()
        for _ in range(5): s,i=ix.search(qd,k)
        torch.cuda.synchronize(); t1=time.time()
        st={kk:ix.get_stat(kk) for kk in ("last_exact_kx","last_exact_band_queries","last_exact_band_passes","last_overflow")}
    rs,ri=flat_ip_topk(q.numpy(),x.numpy(),k)
    s=s.cpu().numpy(); i=i.cpu().numpy()
    rec=np.mean([len(set(a)&set(b))/k for a,b in zip(i,ri)])
    same=np.mean((i==ri).all(1))
    print(dt,N,D,nq,k,"ms",(t1-t0)/5*1e3,st,"recall",rec,"rows same order",same,"max|ds|",np.abs(s-rs).max(), flush=True)
    return rec
run(100000,384,32,10,torch.float16)
run(200000,768,256,100,torch.float16)
run(200000,1024,128,200,torch.bfloat16)
run(200000,768,64,100,torch.float16,expand=100)   # k' = k+16: band passes expected
run(50000,128,16,50,torch.bfloat16,expand=100,cap=256)
