"""Poseidon-BN254 over an emulated field: mirror of the reference's
hash/emulated/bn254/poseidon/poseidon.go -- the permutation of hash/native/bn254/poseidon with every
state element an ``emulated.Element[sw_bn254.ScalarField]`` and every operation a call on
``emulated.Field`` (``New`` :27-37, ``Write`` :42-48, ``Sum`` :53-115, ``SumIsEqual`` /
``AssertSumIsEqual`` :123-133, ``sigma`` / ``ark`` / ``mix`` / ``mixLast`` :135-165, ``Hash`` :172,
``MultiHash`` :182-212, ``AssertMultiHashEqual`` :215-226).  The host field may be any curve's scalar
field in gnark (the reference's tests compile for BLS12-377); here it is BN254's own, the only one
this prover has -- the emulated arithmetic does not care.
"""
from ..std import emulated
from .poseidon import MAX_HASH_INPUTS, MAX_MULTIHASH_INPUTS
from .poseidon_constants import N_ROUNDS_F, N_ROUNDS_P, opt_params

ScalarField = emulated.BN254Fr      # sw_bn254.ScalarField


class Poseidon:
    def __init__(self, api, field=None):
        self.api = api
        self.field = field if field is not None else emulated.NewField(api, ScalarField)
        self.data = []

    def Write(self, *data):
        if len(self.data) + len(data) > MAX_HASH_INPUTS:
            return
        self.data.extend(data)

    def Reset(self):
        self.data = []

    def WriteSucceeded(self):
        return len(self.data) > 0

    def Sum(self):
        f = self.field
        t = len(self.data) + 1
        rp = N_ROUNDS_P[t - 2]
        rf2 = N_ROUNDS_F // 2
        c, m, p, s = opt_params(t)
        state = [f.NewElement(0)] + [f.NewElement(x) for x in self.data]
        state = self._ark(state, c, 0)
        for r in range(rf2 - 1):
            state = [self._sigma(x) for x in state]
            state = self._ark(state, c, (r + 1) * t)
            state = self._mix(state, m)
        state = [self._sigma(x) for x in state]
        state = self._ark(state, c, rf2 * t)
        state = self._mix(state, p)
        for r in range(rp):
            state[0] = self._sigma(state[0])
            state[0] = f.Add(state[0], f.NewElement(c[(rf2 + 1) * t + r]))
            base = (2 * t - 1) * r
            new0 = f.Zero()
            for j in range(t):
                new0 = f.Add(new0, f.Mul(f.NewElement(s[base + j]), state[j]))
            for k in range(1, t):
                state[k] = f.Add(state[k], f.Mul(state[0], f.NewElement(s[base + t + k - 1])))
            state[0] = new0
        for r in range(rf2 - 1):
            state = [self._sigma(x) for x in state]
            state = self._ark(state, c, (rf2 + 1) * t + rp + r * t)
            state = self._mix(state, m)
        state = [self._sigma(x) for x in state]
        out = self._mix_last(state, m, 0)
        self.data = []
        return out

    def SumIsEqual(self, expected):
        f = self.field
        return f.IsZero(f.Sub(self.Sum(), expected))

    def AssertSumIsEqual(self, expected):
        self.api.AssertIsEqual(self.SumIsEqual(expected), 1)

    def _sigma(self, x):
        f = self.field
        x2 = f.Mul(x, x)
        x4 = f.Mul(x2, x2)
        return f.Mul(x4, x)

    def _ark(self, state, c, r):
        f = self.field
        return [f.Add(v, f.NewElement(c[i + r])) for i, v in enumerate(state)]

    def _mix(self, state, m):
        f, t = self.field, len(state)
        out = []
        for i in range(t):
            acc = f.Zero()
            for j in range(t):
                acc = f.Add(acc, f.Mul(f.NewElement(m[j][i]), state[j]))
            out.append(acc)
        return out

    def _mix_last(self, state, m, r):
        f = self.field
        acc = f.Zero()
        for j in range(len(state)):
            acc = f.Add(acc, f.Mul(f.NewElement(m[j][r]), state[j]))
        return acc


def New(api):
    return Poseidon(api)


def Hash(api, *inputs):
    h = Poseidon(api)
    h.Write(*inputs)
    return h.Sum()


def MultiHash(api, *inputs):
    n = len(inputs)
    if n <= MAX_HASH_INPUTS:
        return Hash(api, *inputs)
    if n > MAX_MULTIHASH_INPUTS:
        raise ValueError(f"the maximum number of inputs supported is {MAX_MULTIHASH_INPUTS}")
    h = Poseidon(api)
    hashed = []
    for i in range(0, n, MAX_HASH_INPUTS):
        h.Write(*inputs[i:i + MAX_HASH_INPUTS])
        hashed.append(h.Sum())
        h.Reset()
    if len(hashed) == 1:
        return hashed[0]
    if len(hashed) <= MAX_HASH_INPUTS:
        h.Write(*hashed)
        return h.Sum()
    return MultiHash(api, *hashed)


def AssertMultiHashEqual(api, inputs, expected):
    res = MultiHash(api, *inputs)
    emulated.NewField(api, ScalarField).AssertIsEqual(res, expected)
