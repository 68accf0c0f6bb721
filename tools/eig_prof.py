"""The spectral decomposition op: direct solver (eig.hip) beside the Jacobi kernel (cluster.hip) - accuracy against a
float64 eigh and time per call.  Dev tool, GPU only.

    python tools/eig_prof.py [P N K]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerclip_amd import _lib as L
import centerclip_amd.torch_ops  # noqa: F401  (registers torch.ops.centerclip)


def laplacians(P, N, D, scale, sigma, knn, planted, gen):
    if planted:
        W = torch.full((P, N, N), 1e-6, dtype=torch.float64)
        b = N // planted
        for p in range(P):
            for c in range(planted):
                blk = torch.rand(b, b, generator=gen, dtype=torch.float64) * 0.5 + 0.5
                W[p, c * b:(c + 1) * b, c * b:(c + 1) * b] = 0.5 * (blk + blk.T)
    else:
        X = torch.randn(P, N, D, generator=gen, dtype=torch.float64) * scale
        n1 = (X * X).sum(-1, keepdim=True)
        W = torch.exp(-(n1 + n1.transpose(1, 2) - 2 * X @ X.transpose(1, 2)) / (2 * sigma ** 2))
        if knn:
            kth = W.topk(knn, dim=-1).values[..., -1:]
            keep = W >= kth
            W = W * (keep | keep.transpose(1, 2))
    deg = W.sum(-1)
    inv = deg.pow(-0.5)
    return ((torch.diag_embed(deg) - W) * inv[:, :, None] * inv[:, None, :]).float()


def check(tag, Lm, K, jacobi):
    sv = 1 if jacobi else 0                      # CC_EIG_JACOBI / CC_EIG_AUTO
    Ld = Lm.cuda()
    Q, ev, sw = torch.ops.centerclip.spectral_embedding(Ld, K, True, sv)
    torch.cuda.synchronize()
    Qd = Q[:, :, :K].double().cpu(); evd = ev.double().cpu(); L64 = Lm.double()
    ref = torch.linalg.eigvalsh(L64)[:, :K].flip(-1)
    res = float((L64 @ Qd - Qd * evd[:, None, :]).abs().max())
    orth = float((Qd.transpose(1, 2) @ Qd - torch.eye(K, dtype=torch.float64)).abs().max())
    everr = float((evd - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(1): torch.ops.centerclip.spectral_embedding(Ld, K, True, sv)
    e0.record()
    reps = 20 if Lm.shape[1] <= 200 else 3
    for _ in range(reps): torch.ops.centerclip.spectral_embedding(Ld, K, True, sv)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:34s} {'jacobi' if jacobi else 'direct':7s} residual {res:.2e} orth {orth:.2e} eigenvalue err {everr:.2e} "
          f"finite {bool(torch.isfinite(Q).all())}  {e0.elapsed_time(e1) / reps * 1e3:8.1f} us / call", flush=True)


def main():
    gen = torch.Generator().manual_seed(0)
    cases = [("heat 48x196 K49", laplacians(48, 196, 64, 0.25, 2.0, 0, 0, gen), 49),
             ("knn 48x196 K49", laplacians(48, 196, 64, 0.25, 2.0, 10, 0, gen), 49),
             ("heat D768 48x196 K49", laplacians(48, 196, 768, 0.08, 2.0, 0, 0, gen), 49),
             ("planted 49x4 8x196 K49", laplacians(8, 196, 0, 0, 0, 0, 49, gen), 49),
             ("knn 64x147 K49 (12 -> 4 frames)", laplacians(64, 147, 64, 0.25, 2.0, 10, 0, gen), 49),
             ("knn 96x98 K49 (12 -> 6 frames)", laplacians(96, 98, 64, 0.25, 2.0, 10, 0, gen), 49),
             ("heat 16x64 K8", laplacians(16, 64, 32, 0.35, 2.0, 0, 0, gen), 8),
             ("heat 8x100 K25", laplacians(8, 100, 32, 0.35, 2.0, 0, 0, gen), 25),
             ("heat 4x37 K5", laplacians(4, 37, 16, 0.35, 2.0, 0, 0, gen), 5),
             ("heat 4x6 K3", laplacians(4, 6, 8, 0.35, 2.0, 0, 0, gen), 3),
             ("knn 64x392 K49 (cfg 4)", laplacians(64, 392, 64, 0.25, 2.0, 10, 0, gen), 49),
             ("knn 64x588 K100 (cfg 5)", laplacians(64, 588, 64, 0.25, 2.0, 10, 0, gen), 100),
             ("heat 8x230 K12", laplacians(8, 230, 32, 0.3, 2.0, 0, 0, gen), 12),
             ("planted 98x4 4x392 K98", laplacians(4, 392, 0, 0, 0, 0, 98, gen), 98),
             ("heat 2x640 K128", laplacians(2, 640, 64, 0.25, 2.0, 0, 0, gen), 128)]
    if len(sys.argv) > 3:
        P, N, K = (int(a) for a in sys.argv[1:4])
        cases = [(f"heat {P}x{N} K{K}", laplacians(P, N, 64, 0.25, 2.0, 0, 0, gen), K)]
    import ctypes
    prof = torch.zeros(24, dtype=torch.int64, device="cuda")
    names = ["load", "tridiagonalise", "pack", "eigenvalues", "solve", "gram-schmidt", "back-transform", "store"]
    for tag, Lm, K in cases[:1] + cases[6:7]:
        L.lib().cc_debug_set_eig_profile(ctypes.c_void_p(prof.data_ptr()))
        torch.ops.centerclip.spectral_embedding(Lm.cuda(), K, True)
        torch.cuda.synchronize()
        L.lib().cc_debug_set_eig_profile(ctypes.c_void_p(0))
        st = prof.cpu().tolist()
        print(tag, "phases (us, workgroup 0):", ", ".join(f"{n} {(st[i + 1] - st[i]) / 100:.1f}" for i, n in enumerate(names)),
              f"| total {(st[8] - st[0]) / 100:.1f} | tridiagonalise, wave 0: pass {st[16] / 100:.1f}, wait {st[17] / 100:.1f}, scalar part {st[18] / 100:.1f}", flush=True)
    for tag, Lm, K in [c for c in cases if c[1].shape[1] > 196][:2]:
        L.lib().cc_debug_set_eig_profile(ctypes.c_void_p(prof.data_ptr()))
        torch.ops.centerclip.spectral_embedding(Lm.cuda(), K, True)
        torch.cuda.synchronize()
        L.lib().cc_debug_set_eig_profile(ctypes.c_void_p(0))
        st = prof.cpu().tolist()
        nb = ["copy", "tridiagonalise", "eigenvalues", "solve", "gram-schmidt", "back-transform + store"]
        print(tag, "phases (us, workgroup 0):", ", ".join(f"{n} {(st[i + 1] - st[i]) / 100:.0f}" for i, n in enumerate(nb)),
              f"| total {(st[6] - st[0]) / 100:.0f}", flush=True)
    for tag, Lm, K in cases:
        for jac in (False, True):
            check(tag, Lm, K, jac)


if __name__ == "__main__":
    main()
