"""Dev probe (CPU, numpy fp32): the numerics of the direct eigensolver planned for the spectral path - Householder
tridiagonalisation, multi-section Sturm bisection for the K smallest eigenvalues, inverse iteration with one shift per
vector and a Gram-Schmidt pass over ALL K vectors per iteration, back-transformation - on the kinds of L_sym the module
produces (heat kernel, KNN, planted partitions with numerically multiple eigenvalues).  Prints the worst residual,
orthogonality defect and eigenvalue error against a float64 eigh.  Test infrastructure only; nothing imports this.
"""
import sys
import numpy as np

f32 = np.float32
EPS = f32(1.1920929e-7)


def tridiagonalise(A):
    """A symmetric fp32 [N,N] (destroyed) -> d [N], e [N-1], reflectors H [N-2][N] (v with v[k+1] = 1), tau [N-2]."""
    N = A.shape[0]
    A = A.astype(f32).copy()
    H = np.zeros((max(N - 2, 0), N), f32)
    tau = np.zeros(max(N - 2, 0), f32)
    d = np.zeros(N, f32)
    e = np.zeros(N - 1, f32)
    for k in range(N - 2):
        x = A[k, k + 1:].copy()
        x0 = x[0]
        sig = f32(np.dot(x[1:], x[1:]))
        d[k] = A[k, k]
        if sig == 0:
            e[k] = x0
            continue                                             # tau = 0: H_k = I
        nrm = f32(np.sqrt(x0 * x0 + sig))
        beta = -nrm if x0 >= 0 else nrm
        t = f32((beta - x0) / beta)
        s = f32(1.0) / f32(x0 - beta)
        v = (x * s).astype(f32)
        v[0] = 1
        tau[k] = t
        H[k, k + 1:] = v
        e[k] = beta
        S = A[k + 1:, k + 1:]
        p = (t * (S @ v)).astype(f32)
        g = f32(np.dot(p, v))
        w = (p - f32(0.5) * t * g * v).astype(f32)
        S -= np.outer(v, w) + np.outer(w, v)
    d[N - 2] = A[N - 2, N - 2]
    d[N - 1] = A[N - 1, N - 1]
    e[N - 2] = A[N - 2, N - 1]
    return d, e, H, tau


def sturm_count(d, e2, x, pivmin):
    """number of eigenvalues < x, for an array of shifts x (fp32 arithmetic, the kernel's recurrence)."""
    x = np.asarray(x, f32)
    q = d[0] - x
    q = np.where(np.abs(q) < pivmin, -pivmin, q).astype(f32)
    cnt = (q < 0).astype(np.int32)
    for i in range(1, len(d)):
        q = (d[i] - x - e2[i - 1] / q).astype(f32)
        q = np.where(np.abs(q) < pivmin, -pivmin, q).astype(f32)
        cnt += q < 0
    return cnt


def bisect(d, e, K, first=1024, lanes=16, steps=5):
    N = len(d)
    e2 = (e * e).astype(f32)
    ae = np.abs(e)
    r = np.zeros(N, f32); r[:-1] += ae; r[1:] += ae
    lo, hi = f32((d - r).min()), f32((d + r).max())
    span = max(abs(lo), abs(hi))
    lo, hi = f32(lo - 2 * EPS * span * N), f32(hi + 2 * EPS * span * N)
    pivmin = f32(max(1e-30, float(e2.max()) * 1.1754944e-38 / EPS))
    pivmin = f32(max(pivmin, 1e-30))
    xs = (lo + (hi - lo) * (np.arange(1, first + 1, dtype=f32) / f32(first + 1))).astype(f32)
    cnt = sturm_count(d, e2, xs, pivmin)
    los, his = np.full(K, lo, f32), np.full(K, hi, f32)
    for k in range(K):
        below = np.nonzero(cnt <= k)[0]
        above = np.nonzero(cnt > k)[0]
        if len(below): los[k] = xs[below].max()
        if len(above): his[k] = xs[above].min()
    for _ in range(steps):
        frac = (np.arange(1, lanes + 1, dtype=f32) / f32(lanes + 1))
        X = (los[:, None] + (his - los)[:, None] * frac[None, :]).astype(f32)
        C = sturm_count(d, e2, X.reshape(-1), pivmin).reshape(K, lanes)
        for k in range(K):
            b = C[k] <= k
            if b.any(): los[k] = max(los[k], X[k][b].max())
            if (~b).any(): his[k] = min(his[k], X[k][~b].min())
    return ((los + his) * f32(0.5)).astype(f32), pivmin


def solve_shifted(d, e, lam, B, tiny):
    """(T - lam_k) y_k = b_k for every column k, Gaussian elimination with partial pivoting on the tridiagonal, fp32,
    one column = one lane (vectorised over k).  Tiny pivots are replaced (the system is singular on purpose)."""
    N, K = B.shape
    u0 = np.zeros((N, K), f32); u1 = np.zeros((N, K), f32); u2 = np.zeros((N, K), f32)
    y = B.astype(f32).copy()
    a = (d[0] - lam).astype(f32)                                  # current row: (a, b, 0 | rhs r)
    b = np.full(K, e[0], f32) if N > 1 else np.zeros(K, f32)
    c = np.zeros(K, f32)
    r = y[0].copy()
    for i in range(N - 1):
        # next row: (e_i, d_{i+1} - lam, e_{i+1})
        na = np.full(K, e[i], f32)
        nb = (d[i + 1] - lam).astype(f32)
        nc = np.full(K, e[i + 1] if i + 2 < N else 0, f32)
        nr = y[i + 1].copy()
        swap = np.abs(na) > np.abs(a)
        pa = np.where(swap, na, a); pb = np.where(swap, nb, b); pc = np.where(swap, nc, c); pr = np.where(swap, nr, r)
        qa = np.where(swap, a, na); qb = np.where(swap, b, nb); qc = np.where(swap, c, nc); qr = np.where(swap, r, nr)
        pa = np.where(np.abs(pa) < tiny, np.where(pa < 0, -tiny, tiny), pa).astype(f32)
        m = (qa / pa).astype(f32)
        u0[i], u1[i], u2[i], y[i] = pa, pb, pc, pr
        a = (qb - m * pb).astype(f32)
        b = (qc - m * pc).astype(f32)
        c = np.zeros(K, f32)
        r = (qr - m * pr).astype(f32)
    a = np.where(np.abs(a) < tiny, np.where(a < 0, -tiny, tiny), a).astype(f32)
    u0[N - 1] = a; y[N - 1] = r
    x = np.zeros((N, K), f32)
    x[N - 1] = y[N - 1] / u0[N - 1]
    if N > 1:
        x[N - 2] = (y[N - 2] - u1[N - 2] * x[N - 1]) / u0[N - 2]
    for i in range(N - 3, -1, -1):
        x[i] = ((y[i] - u1[i] * x[i + 1] - u2[i] * x[i + 2]) / u0[i]).astype(f32)
    return x


def mgs(Y):
    """right-looking modified Gram-Schmidt over the columns, in order, fp32."""
    Y = Y.astype(f32).copy()
    K = Y.shape[1]
    for k in range(K):
        n = f32(np.sqrt(np.dot(Y[:, k], Y[:, k])))
        Y[:, k] = Y[:, k] / n
        if k + 1 < K:
            c = (Y[:, k] @ Y[:, k + 1:]).astype(f32)
            Y[:, k + 1:] -= np.outer(Y[:, k], c).astype(f32)
    return Y


def start_vectors(N, K):
    """the kernel's start vectors: an integer hash of (row, column) mapped to [-1, 1)."""
    i = np.arange(N, dtype=np.uint64)[:, None]; k = np.arange(K, dtype=np.uint64)[None, :]
    M = np.uint64(0xFFFFFFFF)
    h = (i * np.uint64(0x9E3779B1) + k * np.uint64(0x85EBCA77) + np.uint64(0x165667B1)) & M
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x2C1B3C6D)) & M
    h ^= h >> np.uint64(12); h = (h * np.uint64(0x297A2D39)) & M
    h ^= h >> np.uint64(15)
    return ((h >> np.uint64(8)).astype(f32) * f32(1.0 / 8388608.0) - f32(1.0)).astype(f32)


def inverse_iteration(d, e, lam, iters=3, seed=1):
    N, K = len(d), len(lam)
    Y = start_vectors(N, K)
    nrm = f32(max(np.abs(d).max(), np.abs(e).max()))
    tiny = f32(EPS * nrm)
    # shifts of numerically equal eigenvalues are spread by a few ulps of |T| (LAPACK sstein does the same): equal shifts
    # would make every lane of a cluster converge to the same dominant direction
    sh = lam.astype(f32).copy()
    sep = f32(10) * EPS * nrm
    for k in range(1, K):
        if sh[k] - sh[k - 1] < sep: sh[k] = sh[k - 1] + sep
    for _ in range(iters):
        Y = solve_shifted(d, e, sh, Y, tiny)
        sc = np.abs(Y).max(axis=0)
        Y = (Y / sc).astype(f32)
        Y = mgs(Y)
    return Y


def back_transform(H, tau, Y):
    Z = Y.astype(f32).copy()
    for k in range(H.shape[0] - 1, -1, -1):
        if tau[k] == 0: continue
        v = H[k]
        s = (v @ Z).astype(f32)
        Z -= np.outer(v, tau[k] * s).astype(f32)
    return Z


def smallest_eigenpairs(L, K, iters=3):
    d, e, H, tau = tridiagonalise(L)
    lam, _ = bisect(d, e, K)
    Y = inverse_iteration(d, e, lam, iters)
    Z = back_transform(H, tau, Y)
    return lam, Z


def report(name, Ls, K, iters=3):
    worst = [0, 0, 0, 0]
    for L in Ls:
        L = L.astype(f32)
        lam, Z = smallest_eigenpairs(L, K, iters)
        Ld = L.astype(np.float64); Zd = Z.astype(np.float64)
        ref, V = np.linalg.eigh(Ld)
        res = np.abs(Ld @ Zd - Zd * lam[None, :].astype(np.float64)).max()
        orth = np.abs(Zd.T @ Zd - np.eye(K)).max()
        everr = np.abs(lam.astype(np.float64) - ref[:K]).max()
        gap = ref[K] - ref[K - 1] if K < len(ref) else 1.0
        proj = np.abs(Zd @ Zd.T - V[:, :K] @ V[:, :K].T).max() if gap > 1e-3 else 0.0
        worst = [max(worst[0], res), max(worst[1], orth), max(worst[2], everr), max(worst[3], proj)]
    print(f"{name:40s} residual {worst[0]:.2e}  orth {worst[1]:.2e}  eigenvalue err {worst[2]:.2e}  projector {worst[3]:.2e}")


def heat_lsym(X, sigma):
    n1 = (X * X).sum(-1, keepdims=True)
    d2 = n1 + n1.T - 2 * X @ X.T
    W = np.exp(-d2 / (2 * sigma ** 2))
    return W


def lsym(W):
    deg = W.sum(-1)
    inv = deg ** -0.5
    return ((np.diag(deg) - W) * inv[:, None] * inv[None, :]).astype(f32)


def knn(W, k, mutual=False):
    kth = np.sort(W, axis=-1)[:, -k][:, None]
    keep = W >= kth
    keep = (keep & keep.T) if mutual else (keep | keep.T)
    return W * keep


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rng = np.random.default_rng(0)
    g = np.load("tests/golden/spectral_golden.npz")
    report("fixture knn_lsym (48, K=6)", list(g["knn_lsym"]), 6, iters)
    report("fixture knn_graph_lsym (48, K=6)", list(g["knn_graph_lsym"]), 6, iters)
    mats = [lsym(heat_lsym(rng.standard_normal((196, 64)).astype(f32) * 0.25, 2.0)) for _ in range(3)]
    report("heat kernel N=196 K=49", mats, 49, iters)
    mats = [lsym(knn(heat_lsym(rng.standard_normal((196, 64)).astype(f32) * 0.25, 2.0), 10)) for _ in range(3)]
    report("KNN N=196 K=49", mats, 49, iters)
    mats = [lsym(heat_lsym(rng.standard_normal((196, 768)).astype(f32) * 0.08, 2.0)) for _ in range(2)]
    report("heat kernel N=196 D=768 K=49", mats, 49, iters)
    # planted partitions: K disconnected (or nearly) blocks -> eigenvalue 0 of multiplicity K
    for leak in (0.0, 1e-6, 1e-3):
        mats = []
        for _ in range(2):
            W = np.full((196, 196), leak, np.float64)
            for b in range(49):
                blk = rng.uniform(0.5, 1.0, (4, 4)); blk = 0.5 * (blk + blk.T)
                W[4 * b:4 * b + 4, 4 * b:4 * b + 4] = blk
            perm = rng.permutation(196)
            mats.append(lsym(W[perm][:, perm]))
        report(f"planted 49 x 4, leak {leak:g}", mats, 49, iters)
    mats = [lsym(heat_lsym(rng.standard_normal((64, 32)).astype(f32) * 0.35, 2.0)) for _ in range(3)]
    report("heat kernel N=64 K=8", mats, 8, iters)
    mats = [np.eye(20, dtype=f32), np.diag(np.arange(20)).astype(f32) / 10]
    report("diagonal N=20 K=5", mats, 5, iters)
