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


# ---- processor (circomlib smtprocessor): tree/smt/processor.go:10-72, processor_level.go:10-27,
# processor_sm.go:7-17 -----------------------------------------------------------------------------
def ProcessorSM(api, xor, is0, lev_ins, fnc0, prev_top, prev_old0, prev_bot, prev_new1, prev_na,
                prev_upd):
    aux1 = api.Mul(prev_top, lev_ins)
    aux2 = api.Mul(aux1, fnc0)
    st_top = api.Sub(prev_top, aux1)
    st_old0 = api.Mul(aux2, is0)
    st_new1 = api.Mul(api.Add(api.Sub(aux2, st_old0), prev_bot), xor)
    st_bot = api.Mul(api.Sub(1, xor), api.Add(api.Sub(aux2, st_old0), prev_bot))
    st_upd = api.Sub(aux1, aux2)
    st_na = api.Add(api.Add(api.Add(prev_new1, prev_old0), prev_na), prev_upd)
    return st_top, st_old0, st_bot, st_new1, st_na, st_upd


def ProcessorLevel(api, hFn, st_top, st_old0, st_bot, st_new1, st_upd, sibling, old1leaf, new1leaf,
                   newlrbit, old_child, new_child):
    old_l, old_r = Switcher(api, newlrbit, old_child, sibling)
    old_proof_hash = Hash2(api, hFn, old_l, old_r)
    old_root = api.Add(api.Mul(old1leaf, api.Add(st_bot, st_new1, st_upd)),
                       api.Mul(old_proof_hash, st_top))
    new_l, new_r = Switcher(
        api, newlrbit,
        api.Add(api.Mul(new_child, api.Add(st_top, st_bot)), api.Mul(new1leaf, st_new1)),
        api.Add(api.Mul(sibling, st_top), api.Mul(old1leaf, st_new1)))
    new_proof_hash = Hash2(api, hFn, new_l, new_r)
    new_root = api.Add(api.Mul(new_proof_hash, api.Add(st_top, st_bot, st_new1)),
                       api.Mul(new1leaf, api.Add(st_old0, st_upd)))
    return old_root, new_root


def ProcessorWithLeafHash(api, hFn, old_root, siblings, old_key, hash1_old, is_old0, new_key,
                          hash1_new, fnc0, fnc1):
    api.AssertIsBoolean(is_old0)
    levels = len(siblings)
    enabled = api.Sub(api.Add(fnc0, fnc1), api.Mul(fnc0, fnc1))
    n2b_old = lowBits(api, old_key, levels)
    n2b_new = lowBits(api, new_key, levels)
    smt_lev_ins = LevIns(api, enabled, siblings)
    xors = [api.Xor(n2b_old[i], n2b_new[i]) for i in range(levels)]
    st = [None] * levels
    for i in range(levels):
        prev = (enabled, 0, 0, 0, api.Sub(1, enabled), 0) if i == 0 else st[i - 1]
        st[i] = ProcessorSM(api, xors[i], is_old0, smt_lev_ins[i], fnc0, *prev)
    top, old0, bot, new1, na, upd = st[levels - 1]
    api.AssertIsEqual(api.Add(na, new1, old0, upd), 1)
    lv_old, lv_new = [None] * levels, [None] * levels
    for i in range(levels - 1, -1, -1):
        top, old0, bot, new1, na, upd = st[i]
        oc, nc = (0, 0) if i == levels - 1 else (lv_old[i + 1], lv_new[i + 1])
        lv_old[i], lv_new[i] = ProcessorLevel(api, hFn, top, old0, bot, new1, upd, siblings[i],
                                              hash1_old, hash1_new, n2b_new[i], oc, nc)
    top_l, top_r = Switcher(api, api.Mul(fnc0, fnc1), lv_old[0], lv_new[0])
    ForceEqualIfEnabled(api, old_root, top_l, enabled)
    new_root = api.Add(api.Mul(enabled, api.Sub(top_r, old_root)), old_root)
    are_key_equals = IsEqual(api, old_key, new_key)
    keys_ok = MultiAnd(api, [api.Sub(1, fnc0), fnc1, api.Sub(1, are_key_equals)])
    api.AssertIsEqual(keys_ok, 0)
    return new_root


def Processor(api, hFn, old_root, siblings, old_key, old_value, is_old0, new_key, new_value, fnc0,
              fnc1):
    hash1_old = Hash1(api, hFn, old_key, old_value)
    hash1_new = Hash1(api, hFn, new_key, new_value)
    return ProcessorWithLeafHash(api, hFn, old_root, siblings, old_key, hash1_old, is_old0,
                                 new_key, hash1_new, fnc0, fnc1)
