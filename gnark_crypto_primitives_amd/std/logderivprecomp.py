"""gnark ``std/internal/logderivprecomp``: a two-input function on small integers as a precomputed
lookup table [UPSTREAM-RECALL].  ``uints.New`` builds three of them (XOR, AND, OR on bytes); the
reference reaches them through ``uints.New[uints.U64]`` (utils/uints.go:14-28) and the byte-wise
Keccak of gnark's std/hash/sha3.

The table has one row per input pair, packed into one field element: x | y << bx | f(x, y) << (bx + by).
``Query(x, y)`` takes f(x, y) from a hint (OP_BXOR / OP_BAND in the witness program), packs the
triple the same way and looks it up; all queries of a table share one log-derivative argument
(std/logderivarg.py), built after the circuit's Define.  Inputs must already be range-checked to
their widths (uints' bytes are)."""
from ..frontend.api import OP_BAND, OP_BXOR
from . import logderivarg

_FN = {OP_BXOR: lambda a, b: a ^ b, OP_BAND: lambda a, b: a & b}


class Precomputed:
    def __init__(self, api, op, bits=(8, 8)):
        self.api, self.op, self.bx, self.by = api, op, bits[0], bits[1]
        self.queries, self.rows = [], []
        api.Defer(self._build)

    def Query(self, x, y):
        api = self.api
        res = api.NewHintByteOp(self.op, x, y)
        row = api.Add(x, api.Mul(y, 1 << self.bx))
        self.rows.append(row)
        self.queries.append(api.Add(row, api.Mul(res, 1 << (self.bx + self.by))))
        return res

    def _build(self, api):
        if not self.queries:
            return
        f, bx, by = _FN[self.op], self.bx, self.by
        table = [j | f(j & ((1 << bx) - 1), j >> bx) << (bx + by) for j in range(1 << (bx + by))]
        logderivarg.Build(api, table, self.queries, self.rows)


def New(api, op, bits=(8, 8)):
    """one table per (builder, function, widths)"""
    tabs = api.__dict__.setdefault("_precomputed_tables", {})
    key = (op, tuple(bits))
    if key not in tabs:
        tabs[key] = Precomputed(api, op, bits)
    return tabs[key]
