"""gnark ``std/internal/logderivarg``: log-derivative lookup argument [UPSTREAM-RECALL].

``Build(api, table, queries)`` proves that every query row occurs in the table:
    sum_j m_j / (r - t_j)  ==  sum_i 1 / (r - q_i)
with m_j the multiplicity of table row j among the queries (an unconstrained hint) and r a
challenge from a commitment to the table's variable entries, the queries and the multiplicities
(std/multicommit).  Rows of several columns are folded with powers of the challenge.  Cost: one
constraint per table row (a division), one per query (an inversion), plus the commitment.

This build supports the table shape the range checker needs -- one column holding the constants
0 .. n - 1 -- whose multiplicities the solver counts with one histogram instruction (OP_HIST)."""
from . import multicommit


def BuildRange(api, table_size, queries):
    """table = [0, 1, ..., table_size - 1] (constants), one-column queries."""
    queries = list(queries)
    if not queries:
        return
    mults = api.NewHintCount(queries, table_size)

    def cb(api, challenge):
        lp = 0
        for j, m in enumerate(mults):
            lp = api.Add(lp, api.DivUnchecked(m, api.Sub(challenge, j)))
        rp = 0
        for q in queries:
            rp = api.Add(rp, api.Inverse(api.Sub(challenge, q)))
        api.AssertIsEqual(lp, rp)

    multicommit.WithCommitment(api, cb, *queries, *mults)
