"""Off-circuit BabyJubJub in gnark's reduced twisted-Edwards form (a = -1).

The reference takes these parameters from gnark-crypto (`edbn254.GetEdwardsCurve()`,
elgamal/mul.go:31-36) and iden3's babyjub for test data (elgamal/ciphertext_test.go:178-205).
Parameters cross-checked against the reference's only hard-coded vector.
"""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
A = R - 1
D = 12181644023421730124874158521699555681764249180949974110617291017600649128846
BASE = (9671717474070082183213120605117400219616337014328744928644933853176787189663,
        16950150798460657717958625567821834550301663161624707787222815936182638968203)
ORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041
COFACTOR = 8
IDENTITY = (0, 1)


def _inv(x):
    return pow(x % R, R - 2, R)


def on_curve(p):
    x, y = p
    return (A * x * x + y * y - 1 - D * x * x * y * y) % R == 0


def add(p, q):
    (x1, y1), (x2, y2) = p, q
    k = D * x1 * x2 % R * y1 % R * y2 % R
    return ((x1 * y2 + y1 * x2) * _inv(1 + k) % R, (y1 * y2 - A * x1 * x2) * _inv(1 - k) % R)


def neg(p):
    return (-p[0] % R, p[1])


def mul(p, k):
    acc = IDENTITY
    k = int(k)
    while k:
        if k & 1:
            acc = add(acc, p)
        p = add(p, p)
        k >>= 1
    return acc
