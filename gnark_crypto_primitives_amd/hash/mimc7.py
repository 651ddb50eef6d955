"""iden3 MiMC7 gadget over the native field: mirror of the reference's
hash/native/bn254/mimc7/mimc.go (``New`` :22-29, ``Write`` :34-39 -- silently drops writes beyond 62
inputs --, ``Reset`` :42-45, ``Sum`` :49-56, ``WriteSucceeded`` :61, ``AssertSumIsEqual`` /
``SumIsEqual`` :66-75, ``pow7`` :77-81, ``encrypt`` :83-89); 4 constraints per round, 364 per input.
"""
from .mimc7_native import N_ROUNDS, constants

MAX_INPUTS = 62      # mimc.go:10


class MiMC:
    """hash.Hash[frontend.Variable] (reference hash/hash.go:9-18)."""

    def __init__(self, api):
        self.api = api
        self.params = constants()
        self.h = 0
        self.data = []

    def Write(self, *data):
        if len(self.data) + len(data) > MAX_INPUTS:
            return
        self.data.extend(data)

    def Reset(self):
        self.data = []
        self.h = 0

    def Sum(self):
        for stream in self.data:
            r = self._encrypt(stream)
            self.h = self.api.Add(self.h, r, stream)
        self.data = []
        return self.h

    def WriteSucceeded(self):
        return len(self.data) > 0

    def SumIsEqual(self, expected):
        return self.api.IsZero(self.api.Sub(self.Sum(), expected))

    def AssertSumIsEqual(self, expected):
        self.api.AssertIsEqual(self.SumIsEqual(expected), 1)

    def _pow7(self, x):
        api = self.api
        x2 = api.Mul(x, x)
        x4 = api.Mul(x2, x2)
        return api.Mul(x, x2, x4)

    def _encrypt(self, m):
        x = m
        for i in range(N_ROUNDS):
            x = self._pow7(self.api.Add(x, self.h, self.params[i]))
        return self.api.Add(x, self.h)


def New(api):
    return MiMC(api)
