"""Sparse-Merkle-tree verifier gadget (circomlib smtverifier port, usable with Arbo trees).

Mirror of the reference's tree/smt package: ``InclusionVerifier`` (verifier.go:29),
``ExclusionVerifier`` (:66), ``Verifier`` (:102), ``VerifierWithLeafHash`` (:129),
``VerifierWithLeafHashFlag`` (:171-242), ``VerifierLevel`` (verifier_level.go:8-17),
``VerifierSM`` (verifier_sm.go:5-14), ``LevIns``/``LevInsFlag``/``sumBits`` (lev_ins.go:16-86),
``Hash1``/``Hash2`` (hash.go:10-27), ``lowBits``/``IsEqual``/``ForceEqualIfEnabledFlag``/
``ForceEqualIfEnabled``/``MultiAnd``/``Switcher`` (utils.go:11-56).
``hFn`` is a ``utils.Hasher`` (utils/hashers.go:10): callable (api, *inputs) -> Variable.
"""


def Hash1(api, hFn, key, *values):
    return hFn(api, key, *values, 1)


def Hash2(api, hFn, l, r):
    return hFn(api, l, r)


def lowBits(api, val, n_bits):
    return api.ToBinary(val, n_bits)


def IsEqual(api, a, b):
    return api.IsZero(api.Sub(a, b))


def ForceEqualIfEnabledFlag(api, a, b, enabled):
    diff_zero = api.IsZero(api.Sub(a, b))
    return api.Select(enabled, diff_zero, 1)


def ForceEqualIfEnabled(api, a, b, enabled):
    api.AssertIsEqual(ForceEqualIfEnabledFlag(api, a, b, enabled), 1)


def MultiAnd(api, ins):
    out = 1
    for x in ins:
        out = api.And(out, x)
    return out


def Switcher(api, sel, l, r):
    out_l = api.Select(sel, r, l)
    out_r = api.Select(sel, l, r)
    return out_l, out_r


def sumBits(api, bits):
    acc = 0
    for b in bits:
        acc = api.Add(acc, b)
    return acc


def LevInsFlag(api, enabled, siblings):
    n = len(siblings)
    lev_ins = [None] * n
    if n < 2:
        return api.Select(enabled, 0, 1), lev_ins
    is_zero = [api.IsZero(s) for s in siblings]
    done = [None] * (n - 1)
    lev_ins[n - 1] = api.Sub(1, is_zero[n - 2])
    done[n - 2] = lev_ins[n - 1]
    for i in range(n - 2, 0, -1):
        lev_ins[i] = api.Mul(api.Sub(1, done[i]), api.Sub(1, is_zero[i - 1]))
        done[i - 1] = api.Add(lev_ins[i], done[i])
    lev_ins[0] = api.Sub(1, done[0])
    leaf_zero_ok = is_zero[n - 1]
    one_hot = api.IsZero(api.Sub(sumBits(api, lev_ins), 1))
    valid = api.Select(enabled, api.And(leaf_zero_ok, one_hot), 1)
    return valid, lev_ins


def LevIns(api, enabled, siblings):
    valid, lev_ins = LevInsFlag(api, enabled, siblings)
    api.AssertIsEqual(valid, 1)
    return lev_ins


def VerifierSM(api, is0, lev_ins, fnc, prev_top, prev_i0, prev_iold, prev_inew, prev_na):
    aux1 = api.Mul(prev_top, lev_ins)
    aux2 = api.Mul(aux1, fnc)
    st_top = api.Sub(prev_top, aux1)
    st_inew = api.Sub(aux1, aux2)
    st_iold = api.Mul(aux2, api.Sub(1, is0))
    st_i0 = api.Mul(aux1, is0)
    st_na = api.Add(prev_na, prev_inew, prev_iold, prev_i0)
    return st_top, st_i0, st_iold, st_inew, st_na


def VerifierLevel(api, hFn, st_top, st_iold, st_inew, sibling, old1leaf, new1leaf, lrbit, child):
    proof_l, proof_r = Switcher(api, lrbit, child, sibling)
    proof_hash = Hash2(api, hFn, proof_l, proof_r)
    return api.Add(api.Mul(proof_hash, st_top), api.Mul(old1leaf, st_iold),
                   api.Mul(new1leaf, st_inew))


def VerifierWithLeafHashFlag(api, hFn, enabled, root, siblings, old_key, hash1_old, is_old0,
                             key, hash1_new, fnc):
    n = len(siblings)
    n2b_new = lowBits(api, key, n)
    flag_lev_ins, smt_lev_ins = LevInsFlag(api, enabled, siblings)
    st_top, st_i0, st_iold, st_inew, st_na = ([None] * n for _ in range(5))
    for i in range(n):
        if i == 0:
            prev = (enabled, 0, 0, 0, api.Sub(1, enabled))
        else:
            prev = (st_top[i - 1], st_i0[i - 1], st_iold[i - 1], st_inew[i - 1], st_na[i - 1])
        st_top[i], st_i0[i], st_iold[i], st_inew[i], st_na[i] = VerifierSM(
            api, is_old0, smt_lev_ins[i], fnc, *prev)
    sum_states = api.Add(api.Add(api.Add(st_na[n - 1], st_iold[n - 1]), st_inew[n - 1]),
                         st_i0[n - 1])
    flag_states = IsEqual(api, sum_states, 1)
    levels = [None] * n
    for i in range(n - 1, -1, -1):
        nxt = levels[i + 1] if i < n - 1 else 0
        levels[i] = VerifierLevel(api, hFn, st_top[i], st_iold[i], st_inew[i], siblings[i],
                                  hash1_old, hash1_new, n2b_new[i], nxt)
    are_keys_equal = IsEqual(api, old_key, key)
    key_reuse_ok = MultiAnd(api, [fnc, api.Sub(1, is_old0), are_keys_equal, enabled])
    flag_key_reuse = IsEqual(api, key_reuse_ok, 0)
    flag_root = ForceEqualIfEnabledFlag(api, levels[0], root, enabled)
    return MultiAnd(api, [flag_states, flag_key_reuse, flag_root, flag_lev_ins])


def VerifierWithLeafHash(api, hFn, enabled, root, siblings, old_key, hash1_old, is_old0, key,
                         hash1_new, fnc):
    valid = VerifierWithLeafHashFlag(api, hFn, enabled, root, siblings, old_key, hash1_old,
                                     is_old0, key, hash1_new, fnc)
    api.AssertIsEqual(valid, 1)


def Verifier(api, hFn, enabled, root, siblings, old_key, old_value, is_old0, key, value, fnc):
    hash1_old = Hash1(api, hFn, old_key, old_value)
    hash1_new = Hash1(api, hFn, key, value)
    return VerifierWithLeafHashFlag(api, hFn, enabled, root, siblings, old_key, hash1_old,
                                    is_old0, key, hash1_new, fnc)


def InclusionVerifier(api, hFn, root, siblings, key, value):
    return Verifier(api, hFn, 1, root, siblings, key, value, 0, key, value, 0)


def ExclusionVerifier(api, hFn, root, siblings, old_key, old_value, is_old0, key):
    return Verifier(api, hFn, 1, root, siblings, old_key, old_value, is_old0, key, 0, 1)
