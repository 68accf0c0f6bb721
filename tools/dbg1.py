import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cluster_oracle as co
from oracle.recipes import lattice
import centerclip_amd.cluster as cl
X = lattice(5, (3, 70, 96)); Xd = torch.from_numpy(X).cuda()
for an, sn in ((False, False), (True, True)):
    d = cl.pairwise_distance(Xd, Xd, metric="euclidean", self_nearest=sn, all_negative=an).cpu().numpy()
    ref = co.literal_pairwise_distance(torch.from_numpy(X), torch.from_numpy(X), "euclidean", sn, an, 2.0).numpy()
    bad = np.argwhere(d != ref)
    print(an, sn, "mismatch", len(bad), "of", d.size, "maxabs", np.abs(d-ref).max())
    for b in bad[:5]:
        print(b, d[tuple(b)], ref[tuple(b)], d[tuple(b)]**2 if not an else '')
# sqrt check
v = torch.arange(1, 20000, device="cuda", dtype=torch.float32)
print("torch gpu sqrt vs cpu mismatches:", int((v.sqrt().cpu() != v.cpu().sqrt()).sum()))
