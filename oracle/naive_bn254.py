"""oracle/naive_bn254.py -- definition-level BN254 G1 arithmetic and MSM on Python ints (TEST
INFRASTRUCTURE).  Affine addition with modular inverses, double-and-add; no shared code with the HIP
kernel (Jacobian/Montgomery) or with the service's point encoder.  PARITY UNPINNED w.r.t. the
external prover; the curve itself is the public alt_bn128 (y^2 = x^3 + 3, generator (1,2))."""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
G = (1, 2)


def add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], Q - 2, Q) % Q
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], Q - 2, Q) % Q
    x = (lam * lam - p[0] - q[0]) % Q
    return (x, (lam * (p[0] - x) - p[1]) % Q)


def mul(p, k):
    acc = None
    while k:
        if k & 1:
            acc = add(acc, p)
        p = add(p, p)
        k >>= 1
    return acc


def msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        if p is None or p == (0, 0):
            continue
        acc = add(acc, mul(p, s % R))
    return acc


def on_curve(p):
    return (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0
