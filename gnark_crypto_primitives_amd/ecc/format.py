"""TE <-> RTE coordinate change for BabyJubJub, native-field versions.

Mirror of the reference's ecc/format/twistededwards.go:17 (scaling factor f), ``FromRTEtoTE``
(:29-36), ``FromTEtoRTE`` (:42-48).  gnark uses the reduced form (a = -1), iden3 the standard
form; x_RTE = x_TE * (-f).  ``FromEmulatedRTEtoTE`` / ``FromEmulatedTEtoRTE`` (:53-75) do the same
on ``emulated.Element[sw_bn254.ScalarField]`` with the reference's hard-coded limb constants (:18-23).
"""
SCALING_FACTOR = 6360561867910373094066688120553762416144456282423235903351243436111059670888
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def FromRTEtoTE(api, x, y):
    neg_f_inv = api.Inverse(api.Neg(SCALING_FACTOR))
    return api.Mul(x, neg_f_inv), y


def FromTEtoRTE(api, x, y):
    return api.Mul(x, api.Neg(SCALING_FACTOR)), y


def te_to_rte_native(x, y):
    return x * (-SCALING_FACTOR) % R, y


def rte_to_te_native(x, y):
    return x * pow(-SCALING_FACTOR % R, R - 2, R) % R, y


# ecc/format/twistededwards.go:18-23: -f and (-f)^-1 mod r as 4 x 64-bit limbs, least significant first
EMULATED_NEG_SCALING_FACTOR = (15521113859322357913, 12938262829174804345, 10076105873221699301,
                               2473702300600416990)
EMULATED_INV_NEG_SCALING_FACTOR = (2444430762821907778, 13992585508913553050, 6869659700585691715,
                                   304596441941759207)


def _const_element(field, limbs):
    return field.NewElement(sum(int(v) << (64 * i) for i, v in enumerate(limbs)))


def FromEmulatedRTEtoTE(api, x, y):
    """twistededwards.go:53-61: xTE = x * (-f)^-1 over the emulated BN254 scalar field."""
    from ..std import emulated
    field = emulated.NewField(api, emulated.BN254Fr)
    return field.Mul(x, _const_element(field, EMULATED_INV_NEG_SCALING_FACTOR)), y


def FromEmulatedTEtoRTE(api, x, y):
    """twistededwards.go:66-75: xRTE = x * (-f) over the emulated BN254 scalar field."""
    from ..std import emulated
    field = emulated.NewField(api, emulated.BN254Fr)
    return field.Mul(x, _const_element(field, EMULATED_NEG_SCALING_FACTOR)), y
