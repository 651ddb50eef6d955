"""Hash gadgets (reference: hash/hash.go:9-18 interface, hash/native/hashes.go:12 constructor)."""
from .poseidon import Poseidon, Hash as PoseidonHash, MultiHash as PoseidonMultiHash  # noqa: F401


def MiMC7(api):
    """hash/native/hashes.go:12-14"""
    from .mimc7 import MiMC
    return MiMC(api)


def EmulatedMiMC7(api):
    """hash/emulated/hashes.go:14-16"""
    from .emulated_mimc7 import MiMC
    return MiMC(api)


def EmulatedPoseidon(api):
    """hash/emulated/hashes.go:20-22"""
    from .emulated_poseidon import Poseidon as EP
    return EP(api)
