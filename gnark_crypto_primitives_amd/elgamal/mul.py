"""Fixed-base scalar multiplication with 4-bit windows and nested Lookup2.

Mirror of the reference's elgamal/mul.go: ``initFixedBaseTable`` (:26-72; 63 windows of 16
entries + one of 4, table[i][v] = [v * 2^(4i)] G, built lazily once) and
``FixedBaseScalarMulBN254`` (:76-166).
"""
import functools

from ..ecc import babyjub_native as bjj
from ..std.twistededwards import Curve, Point

NUM_WINDOWS = 64  # 63 * 4 + 2 = 254 bits


@functools.lru_cache(maxsize=1)
def fixed_base_table():
    table = []
    win_base = bjj.BASE                     # 2^(4i) * G
    for i in range(NUM_WINDOWS):
        entries = 4 if i == NUM_WINDOWS - 1 else 16
        row, acc = [bjj.IDENTITY], bjj.IDENTITY
        for _ in range(1, entries):
            acc = bjj.add(acc, win_base)
            row.append(acc)
        table.append(row)
        for _ in range(4):
            win_base = bjj.add(win_base, win_base)
    return table


def FixedBaseScalarMulBN254(api, scalar):
    table = fixed_base_table()
    curve = Curve(api)
    bits = api.ToBinary(scalar, 254)
    res = None
    for i in range(NUM_WINDOWS):
        tab = table[i]
        if i < NUM_WINDOWS - 1:
            b = bits[4 * i:4 * i + 4]
            xs, ys = [t[0] for t in tab], [t[1] for t in tab]
            px = api.Lookup2(b[2], b[3], *[api.Lookup2(b[0], b[1], *xs[4 * k:4 * k + 4])
                                           for k in range(4)])
            py = api.Lookup2(b[2], b[3], *[api.Lookup2(b[0], b[1], *ys[4 * k:4 * k + 4])
                                           for k in range(4)])
            nib_zero = api.And(api.Sub(1, b[0]), api.Sub(1, b[1]))
            nib_zero = api.And(nib_zero, api.Sub(1, b[2]))
            nib_zero = api.And(nib_zero, api.Sub(1, b[3]))
            contrib = Point(px, py)
            if i == 0:
                res = contrib            # first window initialises the accumulator
            else:
                added = curve.Add(res, contrib)
                res = Point(api.Select(nib_zero, res.X, added.X),
                            api.Select(nib_zero, res.Y, added.Y))
        else:
            b = bits[4 * i:4 * i + 2]
            px = api.Lookup2(b[0], b[1], *[t[0] for t in tab])
            py = api.Lookup2(b[0], b[1], *[t[1] for t in tab])
            nib_zero = api.And(api.Sub(1, b[0]), api.Sub(1, b[1]))
            added = curve.Add(res, Point(px, py))
            res = Point(api.Select(nib_zero, res.X, added.X),
                        api.Select(nib_zero, res.Y, added.Y))
    return res
