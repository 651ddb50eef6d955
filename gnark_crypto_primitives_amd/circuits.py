"""The circuits BASELINE.json's configs name, as the reference's own tests define them."""
from .ecc import eddsa
from .elgamal import Ciphertext, DecryptionProof
from .frontend import Public, Secret
from .hash import poseidon
from .std.twistededwards import Point
from .tree import smt
from .utils import PoseidonHasher, PoseidonMultiHasher


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


class ElGamalAddCircuit:
    """testElGamalAddCircuit (elgamal/ciphertext_test.go:25-37): config 4.  A, B, Sum public."""
    A = Public(4)
    B = Public(4)
    Sum = Public(4)

    def define(self, api):
        ct = lambda v: Ciphertext(Point(v[0], v[1]), Point(v[2], v[3]))
        ct(self.A).Add(api, ct(self.A), ct(self.B)).AssertIsEqual(api, ct(self.Sum))


class ElGamalEncryptCircuit:
    """Encrypt circuit of elgamal/encrypt_test.go:61-86 (config 4b): public key and expected
    ciphertext public, k and m secret."""
    PubKey = Public(2)
    Expected = Public(4)
    K = Secret()
    M = Secret()

    def define(self, api):
        z = Ciphertext().Encrypt(api, Point(*self.PubKey), self.K, self.M)
        e = self.Expected
        z.AssertIsEqual(api, Ciphertext(Point(e[0], e[1]), Point(e[2], e[3])))


class DecryptionProofCircuit:
    """testVerifyDecryptionProofCircuit (elgamal/ciphertext_test.go:274-285)."""
    PubKey = Public(2)
    Ct = Public(4)
    A1 = Public(2)
    A2 = Public(2)
    Z = Public()
    Msg = Secret()

    def define(self, api):
        c = self.Ct
        proof = DecryptionProof(Point(*self.A1), Point(*self.A2), self.Z)
        proof.Verify(api, PoseidonMultiHasher, Point(*self.PubKey),
                     Ciphertext(Point(c[0], c[1]), Point(c[2], c[3])), self.Msg)


class EdDSACircuit:
    """testEdDSAVerifierCircuit shape (ecc/bn254/eddsa/verifier_test.go): Poseidon hasher."""
    A = Public(2)
    R = Public(2)
    S = Public()
    Msg = Public()

    def define(self, api):
        v = eddsa.NewVerifier(api, poseidon.Poseidon(api))
        v.Verify(eddsa.PublicKey(Point(*self.A)), eddsa.Signature(Point(*self.R), self.S),
                 self.Msg)


class AddressCircuit:
    """testAddressCircuit (ecc/secp256k1/ecdsa/address_test.go:21-33): config 5.  The public key's
    coordinates are emulated elements, 4 x 64-bit limbs each, least significant first."""
    Address = Public()
    X = Secret(4)
    Y = Secret(4)

    commit = False
    byte_tables = False

    def define(self, api):
        from .ecc.secp256k1 import DeriveAddress, PublicKey
        from .std.emulated import Element
        addr = DeriveAddress(api, PublicKey(Element(self.X), Element(self.Y)), self.commit,
                             self.byte_tables)
        api.AssertIsEqual(self.Address, addr)


class AddressCircuitCommit(AddressCircuit):
    """The same circuit with the bytes of X and Y range-checked as gnark's ``uints.New`` does it for
    an R1CS builder: log-derivative lookup + Groth16 commitment (std/rangecheck.py).  A proof of it
    carries one Pedersen commitment and its proof of knowledge."""
    commit = True


class AddressCircuitByteTables(AddressCircuit):
    """The address circuit the way gnark compiles ``ecdsa.DeriveAddress``: bytes all the way --
    Keccak-f over uints.U64 lanes with the 2^16-row XOR / AND lookup tables, rotations through range
    checks, one Groth16 commitment behind all of them (std/sha3.py::permute_bytes, std/uints.py,
    std/logderivprecomp.py)."""
    byte_tables = True


class EmulatedPoseidonCircuit:
    """hashCircuit of hash/emulated/bn254/poseidon/poseidon_test.go:17-33: Poseidon of three
    emulated BN254 scalars (4 x 64-bit limbs each) equals the public emulated element -- every field
    operation a product check of std/math/emulated (std/emulated.py), one Groth16 commitment behind
    the range checks and the checks' challenge."""
    Expected = Public(4)
    In0 = Secret(4)
    In1 = Secret(4)
    In2 = Secret(4)

    def define(self, api):
        from .hash import emulated_poseidon as ep
        from .std import emulated
        f = emulated.NewField(api, ep.ScalarField)
        ins = [emulated.Element(v, ep.ScalarField) for v in (self.In0, self.In1, self.In2)]
        f.AssertIsEqual(ep.Hash(api, *ins), emulated.Element(self.Expected, ep.ScalarField))

    @staticmethod
    def assignment(ins, expected=None):
        from .hash import poseidon_native
        from .hash.emulated_poseidon import ScalarField as sf
        from .std.emulated import ValueOf
        e = poseidon_native.hash(list(ins)) if expected is None else expected
        return {"Expected": ValueOf(e, sf), "In0": ValueOf(ins[0], sf), "In1": ValueOf(ins[1], sf),
                "In2": ValueOf(ins[2], sf)}
