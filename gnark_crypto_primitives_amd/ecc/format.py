"""TE <-> RTE coordinate change for BabyJubJub, native-field versions.

Mirror of the reference's ecc/format/twistededwards.go:17 (scaling factor f), ``FromRTEtoTE``
(:29-36), ``FromTEtoRTE`` (:42-48).  gnark uses the reduced form (a = -1), iden3 the standard
form; x_RTE = x_TE * (-f).  The emulated-field variants (:53-75) are out of scope (non-native
host curve).
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
