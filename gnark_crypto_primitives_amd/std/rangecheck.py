"""gnark ``std/rangecheck`` in its commitment-based form (``rangecheck.New(api)`` when the builder
is a ``frontend.Committer``: the R1CS / Groth16 case) [UPSTREAM-RECALL].  The reference uses it
through ``uints.New[uints.U64]`` / ``bf.ValueOf`` (utils/uints.go:14-28 <-
ecc/secp256k1/ecdsa/address.go:14-40).

``Check(v, bits)`` only records the request.  After Define (api.Defer) every recorded value is
split by a hint into limbs of ``w`` bits, recomposition is asserted, a most significant limb that
is narrower than w is also looked up shifted to the top of the range, and all limbs go through one
log-derivative lookup into the table 0 .. 2^w - 1 (std/logderivarg).  w minimises
(number of limbs + 2^w), as gnark's getOptimalBasewidth does."""
from . import logderivarg


class _CommitChecker:
    def __init__(self, api):
        self.collected = []
        api.Defer(self._commit)

    def Check(self, v, bits):
        self.collected.append((v, int(bits)))

    @staticmethod
    def _n_limbs(bits, w):
        return -(-bits // w)

    def optimal_width(self):
        best = None
        for w in range(1, 17):
            n = sum(self._n_limbs(b, w) + (1 if b % w else 0) for _, b in self.collected)
            cost = n + (1 << w)
            if best is None or cost < best[0]:
                best = (cost, w)
        return best[1]

    def _commit(self, api):
        if not self.collected:
            return
        w = self.optimal_width()
        limbs_all = []
        for v, bits in self.collected:
            n = self._n_limbs(bits, w)
            limbs = api.NewHintLimbs(v, w, n)
            composed = 0
            for j, l in enumerate(limbs):
                composed = api.Add(composed, api.Mul(l, 1 << (w * j)))
            api.AssertIsEqual(composed, v)
            limbs_all.extend(limbs)
            diff = n * w - bits
            if diff:                       # the top limb has only w - diff bits
                limbs_all.append(api.Mul(limbs[-1], 1 << diff))
        self.collected = []
        logderivarg.BuildRange(api, 1 << w, limbs_all)


def New(api):
    """rangecheck.New: one checker per builder."""
    rc = getattr(api, "_rangechecker", None)
    if rc is None:
        rc = api._rangechecker = _CommitChecker(api)
    return rc
