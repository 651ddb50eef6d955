"""gnark ``std/multicommit``: one Groth16 commitment shared by every gadget that needs
verifier randomness [UPSTREAM-RECALL; the reference reaches it through ``uints.New`` ->
``rangecheck.New`` (utils/uints.go:14-28)].

``WithCommitment(api, cb, *vars)`` registers the variables and a callback; after the circuit's
Define one ``api.Commit`` over all registered variables yields the root challenge, and the k-th
callback receives its k-th power (k = 1 for the first), so that independent arguments use
independent-looking challenges derived from the same commitment."""


class _MultiCommitter:
    def __init__(self, api):
        self.vars, self.cbs = [], []
        api.Defer(self._commit)

    def _commit(self, api):
        if not self.cbs:
            return
        root = api.Commit(*self.vars)
        cmt = root
        for k, cb in enumerate(self.cbs):
            if k:
                cmt = api.Mul(cmt, root)
            cb(api, cmt)


def WithCommitment(api, cb, *committed_vars):
    mc = getattr(api, "_multicommitter", None)
    if mc is None or mc not in getattr(api, "_multicommitters_open", []):
        mc = _MultiCommitter(api)
        api._multicommitter = mc
        api._multicommitters_open = [mc]
    mc.vars.extend(committed_vars)
    mc.cbs.append(cb)
