"""secp256k1 address derivation (config 5) and the byte helpers of the reference's utils package.

Mirrors ecc/secp256k1/ecdsa/address_test.go:35-60 (random account -> SolvingSucceeded),
utils/uints_test.go:33-82, utils/bytes_test.go:30-42 and utils/utils_test.go:38-47.  The
reference holds no address/Keccak vectors of its own (it draws a random key and asks go-ethereum);
the pins here are public ones: Keccak-256 of "" and "hello", the Ethereum addresses of private
keys 1 and 2, and CPython's SHA3-256 through the shared permutation (oracle/pyref.py).
"""
import hashlib
import random

import pytest

from gnark_crypto_primitives_amd import circuits, utils
from gnark_crypto_primitives_amd.ecc.secp256k1 import native as secp
from gnark_crypto_primitives_amd.frontend import Public, Secret, compile_circuit
from gnark_crypto_primitives_amd.std import emulated, sha3
from gnark_crypto_primitives_amd.std.uints import U8, BinaryField
from oracle import pyref

R = pyref.R
PREFIX = "\x19Ethereum Signed Message:\n"


def solve(cc, assignment):
    wires = cc.run_program(cc.assignment_vector(assignment))[0]
    ok, _ = cc.is_satisfied(wires)
    assert ok == (cc.last_status == 0)      # the program's own checks agree with the R1CS
    return ok, wires


@pytest.fixture(scope="module")
def address_cc():
    return compile_circuit(circuits.AddressCircuit())


def address_assignment(priv):
    pub = pyref.secp256k1_mul(priv)
    return {"Address": pyref.eth_address(pub), "X": emulated.limbs_of(pub[0]),
            "Y": emulated.limbs_of(pub[1])}


def test_oracle_keccak_pins():
    rng = random.Random(7)
    for n in (0, 1, 64, 135, 136, 137, 271, 272, 1000):
        d = bytes(rng.randrange(256) for _ in range(n))
        assert pyref.sha3_256(d) == hashlib.sha3_256(d).digest()
    assert pyref.keccak256(b"").hex() == \
        "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert pyref.keccak256(b"hello").hex() == \
        "1c8aff950685c2ed4bc3174f3472287b56d9517b9c948127319a09a7a36deac8"
    assert pyref.eth_address(pyref.secp256k1_mul(1)) == 0x7E5F4552091A69125D5DFCB7B8C2659029395BDF
    assert pyref.eth_address(pyref.secp256k1_mul(2)) == 0x2B5AD5C4795C026514F8317C7A215E218DCCD6CF


def test_native_matches_oracle():
    """The package's assignment-side Keccak / secp256k1 (tables generated from the specification's
    LFSR and (x, y) walk) against the oracle's (published tables)."""
    assert sha3.RC == pyref._KECCAK_RC
    rng = random.Random(8)
    for n in (0, 5, 64, 136, 300):
        d = bytes(rng.randrange(256) for _ in range(n))
        assert secp.keccak256(d) == pyref.keccak256(d)
    for k in (1, 2, 0xDEADBEEF, rng.randrange(secp.N)):
        assert secp.public_key(k) == pyref.secp256k1_mul(k)
        assert secp.address(secp.public_key(k)) == pyref.eth_address(pyref.secp256k1_mul(k))


def test_address_derivation(address_cc):
    """address_test.go:35-60; one Keccak-f permutation = 24 x 6 400 constraints."""
    cc = address_cc
    assert cc.n_public == 2 and cc.n_secret == 8
    assert 150_000 < cc.n_constraints < 155_000 and cc.domain_log2() == 18
    rng = random.Random(9)
    for priv in (1, 2, rng.randrange(1, secp.N)):
        ok, _ = solve(cc, address_assignment(priv))
        assert ok


def test_address_rejects(address_cc):
    cc = address_cc
    a = address_assignment(5)
    bad = dict(a, Address=(a["Address"] + 1) % R)
    assert not solve(cc, bad)[0]
    a2 = address_assignment(6)
    assert not solve(cc, dict(a, Y=a2["Y"]))[0]
    # a limb that does not fit 64 bits cannot be split into eight bytes
    big = dict(a, X=[a["X"][0] + (1 << 64)] + a["X"][1:])
    assert not solve(cc, big)[0]


def test_keccak_gadget_two_blocks():
    """200 message bytes = two absorbed blocks; digest compared byte by byte."""
    n = 200

    class Circuit:
        Msg = Secret(n)
        Digest = Public(32)

        def define(self, api):
            bf = BinaryField(api)
            h = sha3.NewLegacyKeccak256(api)
            h.Write([bf.ByteValueOf(m) for m in self.Msg])
            assert h.Size() == 32
            for got, want in zip(h.Sum(), self.Digest):
                api.AssertIsEqual(got.Val, want)
    cc = compile_circuit(Circuit())
    rng = random.Random(10)
    msg = bytes(rng.randrange(256) for _ in range(n))
    good = {"Msg": list(msg), "Digest": list(pyref.keccak256(msg))}
    assert solve(cc, good)[0]
    bad = dict(good, Msg=[msg[0] ^ 1] + list(msg[1:]))
    assert not solve(cc, bad)[0]
    assert not solve(cc, dict(good, Msg=[256] + list(msg[1:])))[0]     # not a byte


def test_var_to_u8_roundtrip():
    """utils/uints_test.go:15-42.  The reference runs it over BW6-761 where 256 bits fit the
    field; over BN254 the same gadget is exercised with values below 2^253."""
    class Circuit:
        Input = Secret()

        def define(self, api):
            u8s = utils.VarToU8(api, self.Input)
            assert len(u8s) == 32
            api.AssertIsEqual(self.Input, utils.U8ToVar(api, u8s))
    cc = compile_circuit(Circuit())
    rng = random.Random(11)
    for v in (0, 1, (1 << 253) - 1, rng.getrandbits(253)):
        ok, _ = solve(cc, {"Input": v})
        assert ok


def test_elem_to_u8_roundtrip():
    """utils/uints_test.go:44-82 with the BLS12-377 scalar field's limb layout (4 x 64)."""
    params = emulated.BLS12377Fr

    class Circuit:
        Input = Secret(4)

        def define(self, api):
            u8s = utils.ElemToU8(api, emulated.Element(self.Input, params))
            assert len(u8s) == 32
            elem = utils.U8ToElem(api, u8s, params)
            for a, b in zip(self.Input, elem.Limbs):
                api.AssertIsEqual(a, b)
            # byte order: limb 0's least significant byte first
            api.AssertIsEqual(u8s[0].Val, api.FromBinary(*api.ToBinary(self.Input[0], 64)[:8]))
    cc = compile_circuit(Circuit())
    rng = random.Random(12)
    x = rng.randrange(params.modulus)
    assert solve(cc, {"Input": emulated.limbs_of(x, params)})[0]
    limbs = emulated.limbs_of(x, params)
    assert sum(v << (64 * i) for i, v in enumerate(limbs)) == x


def test_pack_unpack_scalar():
    """utils/utils_test.go:17-47."""
    params = emulated.BLS12377Fr

    class Circuit:
        Input = Secret(4)

        def define(self, api):
            packed = utils.PackScalarToVar(api, emulated.Element(self.Input, params))
            unpacked = utils.UnpackVarToScalar(api, packed, params)
            for a, b in zip(self.Input, unpacked.Limbs):
                api.AssertIsEqual(a, b)
    cc = compile_circuit(Circuit())
    x = random.Random(13).randrange(params.modulus)      # 253 bits: fits BN254's field
    assert solve(cc, {"Input": emulated.limbs_of(x, params)})[0]


def test_prefixed_bytes():
    """utils/bytes_test.go:14-42: constant byte strings (zero-filled at the end, truncated)."""
    prefix = utils.BytesFromString(PREFIX, 26)
    hexs = "d03191e177f9ecdd5230e11686b303bfcf770315fd699f2d1e9c12125fdf40f4" * 4
    content = utils.BytesFromString(hexs, 64)
    expected = utils.BytesFromString(PREFIX + hexs, 26 + 64)
    assert prefix.Values() == list(PREFIX.encode())
    assert content.Values() == list(hexs.encode()[:64])
    assert utils.BytesFromBigInt(0x0102, 4).Values() == [1, 2, 0, 0]

    class Circuit:
        Content = Secret(64)
        Prefix = Secret(26)
        Expected = Secret(90)

        def define(self, api):
            a = utils.Bytes([U8(v) for v in self.Prefix] + [U8(v) for v in self.Content])
            b = utils.Bytes([U8(v) for v in self.Expected])
            a.AssertIsEqual(api, b)
            api.AssertIsEqual(a.IsEqual(api, b), 1)
            api.AssertIsEqual(a.IsEqual(api, utils.Bytes(b[:-1])), 0)
            api.AssertIsEqual(utils.StrictCmp(api, self.Prefix[0], self.Prefix[1]), 1)
            api.AssertIsEqual(utils.StrictCmp(api, self.Prefix[0], self.Expected[0]), 0)
    cc = compile_circuit(Circuit())
    asg = {"Content": content.Values(), "Prefix": prefix.Values(), "Expected": expected.Values()}
    assert solve(cc, asg)[0]
    asg["Expected"] = [expected.Values()[0] ^ 1] + expected.Values()[1:]
    assert not solve(cc, asg)[0]
