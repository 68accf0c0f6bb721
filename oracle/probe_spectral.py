"""Dev-container probe (imports /root/reference): how far the reference's spectral clustering result depends on the
eigensolver's rounding - its own pipeline re-run with a float64 eigh in place of the fp32 svd.
Test infrastructure only; nothing imports this."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/reference/modules")
from cluster.spectral import batch_spectral_clustering, constructW
from cluster.fast_kmeans import batch_fast_kmedoids
torch.manual_seed(0)
def alt(X, K, sigma, mode, knn_k, norm_p, distance='euclidean'):
    W = constructW(X, X, sigma=sigma, mode=mode, knn_k=knn_k).double()
    W = 0.5*(W+W.transpose(1,2))
    d = W.sum(-1); inv = d.pow(-0.5)
    L = torch.diag_embed(d) - W
    Ls = inv[:,:,None]*L*inv[:,None,:]
    lam, V = torch.linalg.eigh(Ls)
    # K smallest |lam|
    idx = lam.abs().argsort(dim=-1)[:, :K]
    Q = torch.gather(V, 2, idx[:,None,:].expand(-1, V.shape[1], -1)).float()
    Q = Q / (Q.norm(p=2, dim=-1, keepdim=True) + 1e-6)
    a, m = batch_fast_kmedoids(Q, K, distance=distance, threshold=1e-6, iter_limit=100, id_sort=True, norm_p=norm_p)
    return a, m, lam
for (P,N,D,K,scale,sigma,mode,p) in [(4,64,32,8,0.35,2.0,'HeatKernel',2.0),(4,64,32,8,0.35,2.0,'KNN',2.0),(4,196,64,49,0.25,2.0,'HeatKernel',2.0),
                                       (4,196,64,49,0.25,2.0,'KNN',2.0),(4,196,64,49,0.25,2.0,'KNN',1.0),(4,64,32,8,0.35,2.0,'HeatKernel',1.0), (2,196,768,49,0.08,2.0,'KNN',2.0)]:
    agree=0; tot=0
    for seed in range(5):
        g=torch.Generator().manual_seed(seed)
        X=torch.randn(P,N,D,generator=g)*scale
        a0,m0=batch_spectral_clustering(X,K,mode=mode,knn_k=10,metric='euclidean',threshold=1e-6,iter_limit=100,norm_p=p,correct_sign=True,split_size=1,sigma=sigma)
        a1,m1,lam=alt(X,K,sigma,mode,10,p)
        agree+=int((m0==m1).all(dim=1).sum()); tot+=P
    print(P,N,D,K,mode,p,'problems with identical medoids:',agree,'/',tot, 'lam[K-1],lam[K]=', float(lam[0].abs().sort().values[K-1]), float(lam[0].abs().sort().values[K]))
