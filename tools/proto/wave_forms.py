"""Index-algebra prototype of the one-wavefront-per-instance kernel (csrc/saip_kernel_wave.hip): every 'register' is a
NumPy array of 64 lane values, v_mfma_f64_16x16x4_f64 is emulated with its documented lane map.  Checks the stored
(B-) form, the product primitive C = S1^T S2 and the blocked symmetric sweep (4 x 4 pivot blocks, fix-ups folded into the
MFMA operands) against numpy.linalg before any of it runs on a GPU."""
import numpy as np

LANE = np.arange(64)
LO, HI = LANE & 15, LANE >> 4


def mfma(a, b, c):
    """D = A B + C, A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, C/D[i][j]: reg r = i // 4... (i = 4 r + hi, j = lo)"""
    A = np.zeros((16, 4)); Bm = np.zeros((4, 16))
    A[LO, HI] = a
    Bm[HI, LO] = b
    D = A @ Bm
    out = [c[r] + D[4 * r + HI, LO] for r in range(4)]
    return out


def store(X, KS, T):
    """B-form: v[ks][t][lane] = X[4 ks + hi][16 t + lo]"""
    return [[X[4 * ks + HI, 16 * t + LO].copy() for t in range(T)] for ks in range(KS)]


def unstore(v, KS, T):
    X = np.zeros((4 * KS, 16 * T))
    for ks in range(KS):
        for t in range(T):
            X[4 * ks + HI, 16 * t + LO] = v[ks][t]
    return X


def mm(A, B, KS, TA, TB):
    """stored(S1^T S2) from stored S1 (4KS x 16TA), S2 (4KS x 16TB): out[4 TA][TB]"""
    out = [[None] * TB for _ in range(4 * TA)]
    for rt in range(TA):
        for ct in range(TB):
            acc = [np.zeros(64) for _ in range(4)]
            for ks in range(KS):
                acc = mfma(A[ks][rt], B[ks][ct], acc)
            for r in range(4):
                out[4 * rt + r][ct] = acc[r]
    return out


def chol4_inv_col(D, col):
    return np.linalg.inv(D)[:, col]


def sweep(v, T, nb):
    """in place: stored symmetric X (16T x 16T) -> stored -X^-1 over the leading 4 nb rows/cols; returns the LDL^T pivots"""
    pivots = []
    for kb in range(nb):
        t0, rho = kb >> 2, 4 * (kb & 3)
        P = [v[kb][t].copy() for t in range(T)]
        # diagonal block: lane (lo = rho + a, hi = b) of P[t0] holds D[b][a]
        D = np.zeros((4, 4))
        for a in range(4):
            for b in range(4):
                D[b, a] = P[t0][(rho + a) + 16 * b]
        L = np.linalg.cholesky(D)
        pivots += list(np.diag(L) ** 2)
        Dinv = np.linalg.inv(D)
        inblk = (LO >= rho) & (LO < rho + 4)
        dinv_lane = np.where(inblk, Dinv[np.clip(LO - rho, 0, 3), HI], 0.0)   # a-operand: rows rho..rho+3 of a 16-row tile
        W = []
        for t in range(T):
            acc = mfma(dinv_lane, P[t], [np.zeros(64) for _ in range(4)])
            W.append(acc[kb & 3])   # W[16 t + lo][hi]
        # column block of the pivot: zero the accumulator entries, operand P' = -delta there
        for ks in range(4 * T):
            v[ks][t0] = np.where(inblk, 0.0, v[ks][t0])
        Pm = [P[t].copy() for t in range(T)]
        Pm[t0] = np.where(inblk, -(HI == LO - rho).astype(float), Pm[t0])
        for rt in range(T):
            for ct in range(T):
                acc = [v[4 * rt + r][ct] for r in range(4)]
                acc = mfma(-W[rt], Pm[ct], acc)
                for r in range(4):
                    v[4 * rt + r][ct] = acc[r]
        for t in range(T):
            v[kb][t] = W[t].copy()
        v[kb][t0] = np.where(inblk, -dinv_lane, v[kb][t0])
    return pivots


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    # products
    S1 = rng.standard_normal((32, 16)); S2 = rng.standard_normal((32, 32))
    C = unstore(mm(store(S1, 8, 1), store(S2, 8, 2), 8, 1, 2), 4, 2)
    print("mm error", np.abs(C - S1.T @ S2).max())
    for T, n in ((2, 30), (2, 18), (1, 6), (2, 32)):
        N = 16 * T
        A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
        X = np.eye(N); X[:n, :n] = A
        v = store(X, 4 * T, T)
        piv = sweep(v, T, (n + 3) // 4)
        R = -unstore(v, 4 * T, T)
        ref = np.linalg.inv(A)
        print(f"sweep T={T} n={n}: inverse error {np.abs(R[:n, :n] - ref).max():.2e}, min pivot {min(piv):.3g}, symmetric {np.abs(R - R.T).max():.1e}")
