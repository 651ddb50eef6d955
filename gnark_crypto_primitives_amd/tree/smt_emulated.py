"""Sparse-Merkle-tree verifier / processor over an emulated field: mirror of the reference's
tree/smt/emulated package -- ``InclusionVerifier`` (verifier.go:10), ``ExclusionVerifier`` (:14),
``Verifier`` (:19-54), ``VerifierLevel`` (verifier_level.go:8-13), ``Processor`` (processor.go:12-67),
``ProcessorLevel`` (processor_level.go:10-27), ``LevIns`` (lev_ins.go:10-29), ``Hash1`` / ``Hash2``
(hash.go:11-20), ``IsEqual`` / ``ForceEqualIfEnabled`` / ``Switcher`` / ``mux2`` / ``mux3``
(utils.go:8-34).  Keys, values, siblings and roots are ``emulated.Element``s, the state machines run
on native flags: the reference takes them from github.com/mdehoog/gnark-circom-smt, here they are
the circomlib state machines of tree/smt.py (``VerifierSM``, ``ProcessorSM``, ``MultiAnd``).
The hash is the emulated Poseidon: the reference imports github.com/mdehoog/poseidon's emulated
circuit (iden3's Poseidon over a generic field); here hash/emulated_poseidon.py on the given field --
``hasher(field, inputs)`` can be replaced (the tests also run the gadgets on a native stand-in
field, which checks their logic against tree/smt.py's witnesses in milliseconds).
"""
from . import smt


def _poseidon(field, inputs):
    from ..hash.emulated_poseidon import Poseidon
    h = Poseidon(field.api, field=field)
    h.Write(*inputs)
    return h.Sum()


def Hash1(field, key, value, hasher=_poseidon):
    return hasher(field, [key, value, field.NewElement(1)])


def Hash2(field, l, r, hasher=_poseidon):
    return hasher(field, [l, r])


def IsEqual(field, a, b):
    return field.IsZero(field.Sub(a, b))


def ForceEqualIfEnabled(field, a, b, enabled):
    field.AssertIsEqual(field.Select(enabled, a, b), b)


def Switcher(field, sel, l, r):
    return field.Select(sel, r, l), field.Select(sel, l, r)


def mux2(api, field, as_, bs, a, b):
    zero = field.Zero()
    return field.Mux(api.FromBinary(as_, bs), zero, a, b, a)


def mux3(api, field, as_, bs, cs, a, b, c):
    zero = field.Zero()
    return field.Mux(api.FromBinary(as_, bs, cs), zero, a, b, a, c, a, b, a)


def LevIns(api, field, enabled, siblings):
    levels = len(siblings)
    if levels < 2:
        raise ValueError("LevIns: at least two levels")
    lev_ins, done = [None] * levels, [None] * (levels - 1)
    is_zero = [field.IsZero(s) for s in siblings]
    api.AssertIsEqual(api.Mul(api.Sub(is_zero[levels - 1], 1), enabled), 0)
    lev_ins[levels - 1] = api.Sub(1, is_zero[levels - 2])
    done[levels - 2] = lev_ins[levels - 1]
    for i in range(levels - 2, 0, -1):
        lev_ins[i] = api.Mul(api.Sub(1, done[i]), api.Sub(1, is_zero[i - 1]))
        done[i - 1] = api.Add(lev_ins[i], done[i])
    lev_ins[0] = api.Sub(1, done[0])
    return lev_ins


def VerifierLevel(api, field, st_top, st_iold, st_inew, sibling, old1leaf, new1leaf, lrbit, child,
                  hasher=_poseidon):
    l, r = Switcher(field, lrbit, child, sibling)
    return mux3(api, field, st_top, st_iold, st_inew, Hash2(field, l, r, hasher), old1leaf, new1leaf)


def Verifier(api, field, enabled, root, siblings, old_key, old_value, is_old0, key, value, fnc,
             hasher=_poseidon):
    n = len(siblings)
    hash1_old = Hash1(field, old_key, old_value, hasher)
    hash1_new = Hash1(field, key, value, hasher)
    n2b_new = field.ToBits(key)
    lev_ins = LevIns(api, field, enabled, siblings)
    st = [None] * n
    for i in range(n):
        prev = (enabled, 0, 0, 0, api.Sub(1, enabled)) if i == 0 else st[i - 1]
        st[i] = smt.VerifierSM(api, is_old0, lev_ins[i], fnc, *prev)      # top, i0, iold, inew, na
    top, i0, iold, inew, na = st[n - 1]
    api.AssertIsEqual(api.Add(api.Add(api.Add(na, iold), inew), i0), 1)
    child = field.Zero()
    for i in range(n - 1, -1, -1):
        child = VerifierLevel(api, field, st[i][0], st[i][2], st[i][3], siblings[i], hash1_old,
                              hash1_new, n2b_new[i], child, hasher)
    keys_ok = smt.MultiAnd(api, [fnc, api.Sub(1, is_old0), IsEqual(field, old_key, key), enabled])
    api.AssertIsEqual(keys_ok, 0)
    ForceEqualIfEnabled(field, child, root, enabled)


def InclusionVerifier(api, field, root, siblings, key, value, hasher=_poseidon):
    Verifier(api, field, 1, root, siblings, key, value, 0, key, value, 0, hasher)


def ExclusionVerifier(api, field, root, siblings, old_key, old_value, is_old0, key, hasher=_poseidon):
    Verifier(api, field, 1, root, siblings, old_key, old_value, is_old0, key, field.Zero(), 1, hasher)


def ProcessorLevel(api, field, st_top, st_old0, st_bot, st_new1, st_upd, sibling, old1leaf, new1leaf,
                   newlrbit, old_child, new_child, hasher=_poseidon):
    ol, or_ = Switcher(field, newlrbit, old_child, sibling)
    old_proof_hash = Hash2(field, ol, or_, hasher)
    old_root = mux2(api, field, api.Add(api.Add(st_bot, st_new1), st_upd), st_top, old1leaf,
                    old_proof_hash)
    a = mux2(api, field, api.Add(st_top, st_bot), st_new1, new_child, new1leaf)
    b = mux2(api, field, st_top, st_new1, sibling, old1leaf)
    nl, nr = Switcher(field, newlrbit, a, b)
    new_proof_hash = Hash2(field, nl, nr, hasher)
    new_root = mux2(api, field, api.Add(api.Add(st_top, st_bot), st_new1), api.Add(st_old0, st_upd),
                    new_proof_hash, new1leaf)
    return old_root, new_root


def Processor(api, field, old_root, siblings, old_key, old_value, is_old0, new_key, new_value, fnc0,
              fnc1, hasher=_poseidon):
    levels = len(siblings)
    enabled = api.Sub(api.Add(fnc0, fnc1), api.Mul(fnc0, fnc1))
    hash1_old = Hash1(field, old_key, old_value, hasher)
    hash1_new = Hash1(field, new_key, new_value, hasher)
    n2b_old, n2b_new = field.ToBits(old_key), field.ToBits(new_key)
    lev_ins = LevIns(api, field, enabled, siblings)
    xors = [api.Xor(n2b_old[i], n2b_new[i]) for i in range(levels)]
    st = [None] * levels
    for i in range(levels):
        prev = (enabled, 0, 0, 0, api.Sub(1, enabled), 0) if i == 0 else st[i - 1]
        st[i] = smt.ProcessorSM(api, xors[i], is_old0, lev_ins[i], fnc0, *prev)
        # top, old0, bot, new1, na, upd
    top, old0, bot, new1, na, upd = st[levels - 1]
    api.AssertIsEqual(api.Add(api.Add(na, new1), api.Add(old0, upd)), 1)
    old_child = new_child = field.Zero()
    for i in range(levels - 1, -1, -1):
        old_child, new_child = ProcessorLevel(api, field, st[i][0], st[i][1], st[i][2], st[i][3],
                                              st[i][5], siblings[i], hash1_old, hash1_new, n2b_new[i],
                                              old_child, new_child, hasher)
    top_l, top_r = Switcher(field, api.Mul(fnc0, fnc1), old_child, new_child)
    ForceEqualIfEnabled(field, old_root, top_l, enabled)
    new_root = field.Select(enabled, top_r, old_root)
    keys_ok = smt.MultiAnd(api, [api.Sub(1, fnc0), fnc1, api.Sub(1, IsEqual(field, old_key, new_key))])
    api.AssertIsEqual(keys_ok, 0)
    return new_root
