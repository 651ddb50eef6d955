"""Poseidon-BN254 gadget, iden3/circomlib-compatible optimised form.

Mirror of the reference's hash/native/bn254/poseidon/poseidon.go: ``Hash`` (:38), ``MultiHash``
(:54-91), ``Poseidon.Write`` (:103, silently drops writes that would exceed 16 inputs),
``Reset`` (:111), ``Sum`` (:116-183), ``SumIsEqual``/``AssertSumIsEqual`` (:189-197),
``sigma``/``ark``/``mix``/``mixLast`` (:199-233).  Same call order on the API so the constraint
system has the same shape (Hash2 = 240 constraints, Hash1(k,v,1) = 258 ... SURVEY.md §8a).
"""
from .poseidon_constants import N_ROUNDS_F, N_ROUNDS_P, opt_params

MAX_MULTIHASH_INPUTS = 4096   # poseidon.go:12
MAX_HASH_INPUTS = 16          # poseidon.go:14


class Poseidon:
    """hash.Hash[frontend.Variable] implementation (reference hash/hash.go:9-18)."""

    def __init__(self, api):
        self.api = api
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
        api = self.api
        t = len(self.data) + 1
        rp = N_ROUNDS_P[t - 2]
        rf2 = N_ROUNDS_F // 2
        c, m, p, s = opt_params(t)
        state = [0] + list(self.data)
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
            state[0] = api.Add(state[0], c[(rf2 + 1) * t + r])
            base = (2 * t - 1) * r
            new0 = api.Add(0, 0, *[api.Mul(s[base + j], state[j]) for j in range(t)])
            for k in range(1, t):
                state[k] = api.Add(state[k], api.Mul(state[0], s[base + t + k - 1]))
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
        return self.api.IsZero(self.api.Sub(self.Sum(), expected))

    def AssertSumIsEqual(self, expected):
        self.api.AssertIsEqual(self.SumIsEqual(expected), 1)

    def _sigma(self, x):
        api = self.api
        x2 = api.Mul(x, x)
        x4 = api.Mul(x2, x2)
        return api.Mul(x4, x)

    def _ark(self, state, c, r):
        return [self.api.Add(v, c[i + r]) for i, v in enumerate(state)]

    def _mix(self, state, m):
        api = self.api
        t = len(state)
        return [api.Add(0, 0, *[api.Mul(m[j][i], state[j]) for j in range(t)]) for i in range(t)]

    def _mix_last(self, state, m, s):
        api = self.api
        return api.Add(0, 0, *[api.Mul(m[j][s], state[j]) for j in range(len(state))])


def Hash(api, *inputs):
    """poseidon.Hash (poseidon.go:38-45): up to 16 inputs; empty input is an error."""
    h = Poseidon(api)
    h.Write(*inputs)
    if not h.data:
        raise ValueError("bad inputs provided")
    return h.Sum()


def MultiHash(api, *inputs):
    """poseidon.MultiHash (poseidon.go:54-91): chunks of 16, recursive."""
    n = len(inputs)
    if n <= MAX_HASH_INPUTS:
        return Hash(api, *inputs)
    if n > MAX_MULTIHASH_INPUTS:
        raise ValueError(f"the maximum number of inputs supported is {MAX_MULTIHASH_INPUTS}")
    hasher = Poseidon(api)
    hashed = []
    for i in range(0, n, MAX_HASH_INPUTS):
        hasher.Write(*inputs[i:i + MAX_HASH_INPUTS])
        hashed.append(hasher.Sum())
        hasher.Reset()
    if len(hashed) == 1:
        return hashed[0]
    if len(hashed) <= MAX_HASH_INPUTS:
        hasher.Write(*hashed)
        return hasher.Sum()
    return MultiHash(api, *hashed)
