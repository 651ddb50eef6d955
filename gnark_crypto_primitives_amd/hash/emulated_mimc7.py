"""iden3 MiMC7 over the emulated BN254 scalar field: mirror of the reference's
hash/emulated/bn254/mimc7/mimc.go (``New`` :23-35, ``Write`` :40-45, ``Reset`` :48-51, ``Sum`` :55-63
-- Add then ModAdd by the field's own modulus --, ``SumIsEqual`` / ``AssertSumIsEqual`` :72-81,
``pow7`` :83-88 as x2, x3 = x2 x, x5 = x2 x3, x2 x5; ``encrypt`` :90-98): 4 emulated products per round.
"""
from ..std import emulated
from .mimc7_native import N_ROUNDS, constants

MAX_INPUTS = 62
ScalarField = emulated.BN254Fr


class MiMC:
    def __init__(self, api):
        self.api = api
        self.field = emulated.NewField(api, ScalarField)
        self.params = [self.field.NewElement(c) for c in constants()]
        self.h = self.field.Zero()
        self.data = []

    def Write(self, *data):
        if len(self.data) + len(data) > MAX_INPUTS:
            return
        self.data.extend(data)

    def Reset(self):
        self.h = self.field.Zero()
        self.data = []

    def Sum(self):
        f = self.field
        for stream in self.data:
            stream = f.NewElement(stream)
            r = self._encrypt(stream)
            self.h = f.Add(self.h, r)
            self.h = f.ModAdd(self.h, stream, f.Modulus())
        self.data = []
        return self.h

    def WriteSucceeded(self):
        return len(self.data) > 0

    def SumIsEqual(self, expected):
        f = self.field
        return f.IsZero(f.Sub(self.Sum(), expected))

    def AssertSumIsEqual(self, expected):
        self.api.AssertIsEqual(self.SumIsEqual(expected), 1)

    def _pow7(self, x):
        f = self.field
        x2 = f.Mul(x, x)
        x3 = f.Mul(x2, x)
        x5 = f.Mul(x2, x3)
        return f.Mul(x2, x5)

    def _encrypt(self, m):
        f = self.field
        x = m
        for i in range(N_ROUNDS):
            x = self._pow7(f.Add(f.Add(x, self.h), self.params[i]))
        return f.Add(x, self.h)


def New(api):
    return MiMC(api)
