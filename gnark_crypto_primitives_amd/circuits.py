"""The circuits BASELINE.json's configs name, as the reference's own tests define them."""
from .frontend import Public, Secret
from .hash import poseidon
from .tree import smt
from .utils import PoseidonHasher


class PoseidonCircuit:
    """testPoseidonCiruit (hash/native/bn254/poseidon/poseidon_test.go:20-32): config 1."""
    Data = Secret()
    Hash = Public()

    def define(self, api):
        api.AssertIsEqual(poseidon.Hash(api, self.Data), self.Hash)


def smt_inclusion_circuit(levels: int):
    """testVerifierBN254 (tree/test/verifier_bn254_test.go:23-34) with Siblings[levels]: config 2
    at levels = 160; the reference's own test uses 64 (verifier_bls12377_test.go:23-27)."""
    class SmtInclusion:
        Root = Secret()
        Key = Secret()
        Value = Secret()
        Siblings = Secret(levels)

        def define(self, api):
            valid = smt.InclusionVerifier(api, PoseidonHasher, self.Root, self.Siblings, self.Key,
                                          self.Value)
            api.AssertIsEqual(valid, 1)
    return SmtInclusion()


def smt_verifier_circuit(levels: int):
    """smt.Verifier with every selector an input (tree/smt/verifier.go:102): config 3 exercises
    fnc = 0 (inclusion) and fnc = 1 (exclusion) through the same constraint system."""
    class SmtVerifier:
        Root = Secret()
        OldKey = Secret()
        OldValue = Secret()
        IsOld0 = Secret()
        Key = Secret()
        Value = Secret()
        Fnc = Secret()
        Siblings = Secret(levels)

        def define(self, api):
            valid = smt.Verifier(api, PoseidonHasher, 1, self.Root, self.Siblings, self.OldKey,
                                 self.OldValue, self.IsOld0, self.Key, self.Value, self.Fnc)
            api.AssertIsEqual(valid, 1)
    return SmtVerifier()
