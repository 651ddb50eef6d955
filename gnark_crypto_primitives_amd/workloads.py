"""Synthetic assignments for the five configurations BASELINE.json names: what the
reference's tests build with go-ethereum / Arbo / gnark-crypto, rebuilt here from the package's own
off-circuit arithmetic.  ``build(name, ...)`` returns (circuit, assignment generator, label)."""
from . import circuits
from .ecc import babyjub_native as bjj
from .ecc.secp256k1 import native as secp
from .hash import poseidon_native
from .std.emulated import limbs_of
from .tree import smt_witness

R = poseidon_native.R
NAMES = ("arbo", "poseidon", "verifier", "elgamal-add", "elgamal-encrypt", "address",
         "address-commit", "address-bytes", "emulated-poseidon")


def _poseidon(rng):
    d = rng.randrange(R)
    return {"Data": d, "Hash": poseidon_native.hash([d])}


def _verifier(levels, populated):
    def gen(rng):
        """alternating inclusion (fnc = 0) and exclusion (fnc = 1) proofs, tree/smt/verifier.go:
        66-81: the excluded key shares the populated prefix of a present leaf and differs above."""
        w = smt_witness.synthetic_inclusion(rng, levels, populated)
        if rng.getrandbits(1):
            return dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Fnc=0)
        mask = (1 << populated) - 1
        other = (w["Key"] & mask) | (((w["Key"] >> populated) ^ 1) << populated)
        return dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Key=other, Value=0, Fnc=1)
    return gen


def _elgamal_pub(rng):
    return bjj.mul(bjj.BASE, rng.randrange(1, bjj.ORDER))


def _elgamal_add(rng):
    pub = _elgamal_pub(rng)

    def enc(m):
        k = rng.randrange(bjj.ORDER)
        return bjj.mul(bjj.BASE, k) + bjj.add(bjj.mul(bjj.BASE, m), bjj.mul(pub, k))
    a, b = enc(rng.getrandbits(20)), enc(rng.getrandbits(20))
    return {"A": list(a), "B": list(b),
            "Sum": list(bjj.add(a[:2], b[:2]) + bjj.add(a[2:], b[2:]))}


def _elgamal_encrypt(rng):
    pub, k, m = _elgamal_pub(rng), rng.randrange(bjj.ORDER), rng.getrandbits(60)
    ex = bjj.mul(bjj.BASE, k) + bjj.add(bjj.mul(bjj.BASE, m), bjj.mul(pub, k))
    return {"PubKey": list(pub), "Expected": list(ex), "K": k, "M": m}


def _address(rng):
    pub = secp.public_key(rng.randrange(1, secp.N))
    return {"Address": secp.address(pub), "X": limbs_of(pub[0]), "Y": limbs_of(pub[1])}


def build(name, levels=160, populated=10):
    if name == "arbo":
        return (circuits.smt_inclusion_circuit(levels),
                lambda rng: smt_witness.synthetic_inclusion(rng, levels, populated),
                f"Arbo SMT inclusion verifier, {levels} levels, Poseidon leaf hash")
    if name == "poseidon":
        return circuits.PoseidonCircuit(), _poseidon, "single Poseidon hash (config 1)"
    if name == "verifier":
        return (circuits.smt_verifier_circuit(levels), _verifier(levels, populated),
                f"circomlib SMT verifier, {levels} levels, inclusion and exclusion (config 3)")
    if name == "elgamal-add":
        return circuits.ElGamalAddCircuit(), _elgamal_add, "ElGamal homomorphic add (config 4)"
    if name == "elgamal-encrypt":
        return (circuits.ElGamalEncryptCircuit(), _elgamal_encrypt,
                "ElGamal encrypt on BabyJubJub (config 4b)")
    if name == "address":
        return (circuits.AddressCircuit(), _address,
                "secp256k1 address derivation, Keccak-256 in R1CS (config 5)")
    if name == "address-commit":
        return (circuits.AddressCircuitCommit(), _address,
                "secp256k1 address derivation, Keccak-256 in R1CS, bytes range-checked through "
                "gnark's commitment-based checker (config 5 with the Groth16 commitment extension)")
    if name == "address-bytes":
        return (circuits.AddressCircuitByteTables(), _address,
                "secp256k1 address derivation the way gnark compiles it: byte-wise Keccak-256 over "
                "uints.U64 with XOR / AND lookup tables, one Groth16 commitment (config 5)")
    if name == "emulated-poseidon":
        return (circuits.EmulatedPoseidonCircuit(),
                lambda rng: circuits.EmulatedPoseidonCircuit.assignment(
                    [rng.randrange(R) for _ in range(3)]),
                "Poseidon of three emulated BN254 scalars (hash/emulated/bn254/poseidon), "
                "std/math/emulated product checks, one Groth16 commitment")
    raise ValueError(f"unknown workload {name!r}; one of {NAMES}")
