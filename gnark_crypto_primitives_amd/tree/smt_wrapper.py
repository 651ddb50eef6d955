"""Off-circuit assignment builder: ``smt.Assignment``, ``smt.Wrapper`` and ``smt.WrapperArbo``
(tree/smt/wrapper.go:12-31, tree/smt/wrapper_arbo.go:13-184).

The reference wraps a live ``arbo.Tree`` on a PebbleDB (third-party modules, not part of the prover
hot path).  What the circuits depend on is restated here as host logic:

* Arbo's byte conventions (``arbo.BigIntToBytes`` / ``BytesToBigInt``: little-endian, keys and
  values padded to the hash length) -- ``big_int_to_bytes`` / ``bytes_to_big_int``;
* the tree shape Arbo and circomlib's SMT share: leaf = H(key, value, 1), node = H(left, right),
  empty = 0, bit i of the key (LSB first) steers level i, a leaf sits at the first level where its
  path is unique -- ``MemTree``, an in-memory tree with Arbo's ``Get`` / ``GenProof`` / ``Add`` /
  ``Update`` behaviour (``Get`` of an absent key returns the leaf found on its path);
* the wrapper itself: ``Proof`` (inclusion when the key exists, exclusion otherwise), ``SetProof``
  / ``Set`` (insert or update) with the "drop the last sibling when the new leaf was pushed down
  next to an existing one" rule (wrapper_arbo.go:170-172) and zero padding to the circuit depth.

[UPSTREAM-RECALL] for Arbo's internals (vocdoni/arbo is not in /root/reference); the assignments
are pinned by the circuits themselves: every record produced here must satisfy ``smt.Verifier`` /
``smt.Processor`` (tests/test_smt_wrapper.py).
"""
from dataclasses import dataclass, field
from typing import List

from ..hash import poseidon_native

R = poseidon_native.R
HASH_LEN = 32          # arbo.HashFunctionPoseidon.Len()


def big_int_to_bytes(blen: int, x: int) -> bytes:
    """arbo.BigIntToBytes: little-endian, right-padded with zeros to ``blen`` bytes."""
    if x < 0:
        raise ValueError("negative value")
    raw = x.to_bytes((x.bit_length() + 7) // 8, "little")
    if len(raw) > blen:
        raise ValueError(f"value does not fit {blen} bytes")
    return raw + bytes(blen - len(raw))


def bytes_to_big_int(b: bytes) -> int:
    """arbo.BytesToBigInt: little-endian; the empty slice is 0."""
    return int.from_bytes(b, "little")


class KeyNotFound(KeyError):
    pass


class MemTree:
    """In-memory sparse Merkle tree with Arbo's observable behaviour (Poseidon hash function)."""

    def __init__(self, max_levels: int, hasher=poseidon_native.hash):
        self.max_levels = max_levels
        self.hash = hasher
        self.nodes = {}        # node hash -> ("leaf", key, value) | ("mid", left, right)
        self.root = 0

    # -- internals
    def _key_int(self, key_bytes: bytes) -> int:
        k = bytes_to_big_int(key_bytes)
        if k >> self.max_levels:
            # arbo keyPathFromKey: len(k) must not exceed ceil(maxLevels / 8) significant bytes
            raise ValueError("key longer than the tree depth")
        return k

    def _leaf(self, k, v):
        h = self.hash([k, v, 1])
        self.nodes[h] = ("leaf", k, v)
        return h

    def _mid(self, l, r):
        h = self.hash([l, r])
        self.nodes[h] = ("mid", l, r)
        return h

    def _walk(self, k):
        """-> (leaf key | None, leaf value | None, siblings root->leaf) along the path of k."""
        cur, sibs = self.root, []
        for lvl in range(self.max_levels + 1):
            if cur == 0:
                return None, None, sibs
            node = self.nodes[cur]
            if node[0] == "leaf":
                return node[1], node[2], sibs
            _, l, r = node
            if (k >> lvl) & 1:
                sibs.append(l)
                cur = r
            else:
                sibs.append(r)
                cur = l
        raise RuntimeError("tree deeper than max_levels")

    def _fold(self, k, cur, sibs):
        for lvl in range(len(sibs) - 1, -1, -1):
            cur = self._mid(sibs[lvl], cur) if (k >> lvl) & 1 else self._mid(cur, sibs[lvl])
        return cur

    # -- arbo.Tree surface used by the wrapper (byte-slice keys and values)
    def Root(self) -> bytes:
        return big_int_to_bytes(HASH_LEN, self.root)

    def Get(self, key_bytes: bytes):
        """(leaf key bytes, leaf value bytes); raises KeyNotFound carrying the leaf found on the
        path (empty slices when the path ends in an empty node), as arbo.Tree.Get does."""
        k = self._key_int(key_bytes)
        lk, lv, _ = self._walk(k)
        if lk == k:
            return big_int_to_bytes(HASH_LEN, lk), big_int_to_bytes(HASH_LEN, lv)
        e = KeyNotFound(k)
        e.leaf = (b"", b"") if lk is None else (big_int_to_bytes(HASH_LEN, lk),
                                                big_int_to_bytes(HASH_LEN, lv))
        raise e

    def GenProof(self, key_bytes: bytes):
        """(leaf key bytes, leaf value bytes, unpacked siblings root->leaf, exists)."""
        k = self._key_int(key_bytes)
        lk, lv, sibs = self._walk(k)
        sib_bytes = [big_int_to_bytes(HASH_LEN, s) for s in sibs]
        if lk is None:
            return b"", b"", sib_bytes, False
        return (big_int_to_bytes(HASH_LEN, lk), big_int_to_bytes(HASH_LEN, lv), sib_bytes,
                lk == k)

    def Add(self, key_bytes: bytes, value_bytes: bytes):
        k, v = self._key_int(key_bytes), bytes_to_big_int(value_bytes) % R
        lk, lv, sibs = self._walk(k)
        if lk == k:
            raise ValueError("key already exists")
        new = self._leaf(k, v)
        if lk is not None:
            # push both leaves down to the first level where their paths differ
            d = len(sibs)
            while d < self.max_levels and ((k >> d) & 1) == ((lk >> d) & 1):
                d += 1
            if d >= self.max_levels:
                raise ValueError("max level reached")
            old = self._leaf(lk, lv)
            sibs = sibs + [0] * (d - len(sibs)) + [old]
        self.root = self._fold(k, new, sibs)

    def Update(self, key_bytes: bytes, value_bytes: bytes):
        k, v = self._key_int(key_bytes), bytes_to_big_int(value_bytes) % R
        lk, _, sibs = self._walk(k)
        if lk != k:
            raise KeyNotFound(k)
        self.root = self._fold(k, self._leaf(k, v), sibs)


@dataclass
class Assignment:
    """smt.Assignment (tree/smt/wrapper.go:20-31)."""
    Fnc0: int = 0
    Fnc1: int = 0
    OldKey: int = 0
    NewKey: int = 0
    IsOld0: int = 0
    OldValue: int = 0
    NewValue: int = 0
    OldRoot: int = 0
    NewRoot: int = 0
    Siblings: List[int] = field(default_factory=list)


class WrapperArbo:
    """smt.WrapperArbo over a ``MemTree`` (tree/smt/wrapper_arbo.go:19-184).  ``SetProof`` leaves
    the tree untouched (the reference discards its write transaction), ``Set`` commits."""

    def __init__(self, tree: MemTree, levels: int):
        self.tree = tree
        self.levels = levels

    def _pad(self, sib_bytes):
        if len(sib_bytes) > self.levels:
            raise ValueError("proof deeper than the circuit")
        out = [bytes_to_big_int(s) for s in sib_bytes]
        return out + [0] * (self.levels - len(out))

    def Proof(self, key: int) -> Assignment:
        """wrapper_arbo.go:31-79: membership (Fnc0 = 0) when the key exists, else
        non-membership (Fnc0 = 1) against the leaf found on its path."""
        t = self.tree
        a = Assignment(NewKey=key)
        a.OldRoot = a.NewRoot = bytes_to_big_int(t.Root())
        ok, ov, sibs, exists = t.GenProof(big_int_to_bytes(HASH_LEN, key))
        if exists:
            a.Fnc0 = 0
            a.NewValue = bytes_to_big_int(ov)
        else:
            a.Fnc0 = 1
        a.OldKey, a.OldValue = bytes_to_big_int(ok), bytes_to_big_int(ov)
        a.IsOld0 = 0 if len(ok) > 0 else 1
        a.Siblings = self._pad(sibs)
        return a

    def _set(self, tree: MemTree, key: int, value: int) -> Assignment:
        a = Assignment(NewKey=key, NewValue=value)
        a.OldRoot = bytes_to_big_int(tree.Root())
        kb, vb = big_int_to_bytes(HASH_LEN, key), big_int_to_bytes(HASH_LEN, value)
        try:
            ok, ov = tree.Get(kb)
            a.Fnc0, a.Fnc1 = 0, 1
            tree.Update(kb, vb)
        except KeyNotFound as e:
            ok, ov = e.leaf
            a.Fnc0, a.Fnc1 = 1, 0
            tree.Add(kb, vb)
        a.OldKey, a.OldValue = bytes_to_big_int(ok), bytes_to_big_int(ov)
        a.IsOld0 = 0 if len(ok) > 0 else 1
        a.NewRoot = bytes_to_big_int(tree.Root())
        _, _, sibs, exists = tree.GenProof(kb)
        if not exists:
            raise KeyError("key not found")
        if a.IsOld0 == 0 and a.Fnc1 == 0:
            sibs = sibs[:-1]          # the pushed-down old leaf: the circuit derives it itself
        a.Siblings = self._pad(sibs)
        return a

    def SetProof(self, key: int, value: int) -> Assignment:
        import copy
        scratch = copy.copy(self.tree)
        scratch.nodes = dict(self.tree.nodes)
        return self._set(scratch, key, value)

    def Set(self, key: int, value: int) -> Assignment:
        return self._set(self.tree, key, value)


def delete_assignment(insert: Assignment) -> Assignment:
    """circomlib smtprocessor fnc = (1, 1): the deletion that undoes ``insert`` (same siblings,
    old = the leaf that stays, new = the leaf that goes, roots swapped).  Arbo has no delete; the
    Processor gadget supports it (tree/smt/processor.go:10-72)."""
    if (insert.Fnc0, insert.Fnc1) != (1, 0):
        raise ValueError("expects an insert assignment")
    return Assignment(1, 1, insert.OldKey, insert.NewKey, insert.IsOld0, insert.OldValue,
                      insert.NewValue, insert.NewRoot, insert.OldRoot, list(insert.Siblings))
