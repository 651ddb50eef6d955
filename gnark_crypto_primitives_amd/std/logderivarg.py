"""gnark ``std/internal/logderivarg``: log-derivative lookup argument [UPSTREAM-RECALL].

``Build(api, table, queries)`` proves that every query row occurs in the table:
    sum_j m_j / (r - t_j)  ==  sum_i 1 / (r - q_i)
with m_j the multiplicity of table row j among the queries (an unconstrained hint) and r a
challenge from a commitment to the table's variable entries, the queries and the multiplicities
(std/multicommit).  Rows of several columns are folded with powers of the challenge.  Cost: one
constraint per table row (a division), one per query (an inversion), plus the commitment.

This build supports one-column tables of constants -- the range checker's 0 .. n - 1 and the packed
tables of std/logderivprecomp.py (x | y << 8 | f(x, y) << 16) -- whose multiplicities the solver
counts with one histogram instruction (OP_HIST) over the queries' row indices."""
from . import multicommit


def BuildRange(api, table_size, queries):
    """table = [0, 1, ..., table_size - 1] (constants), one-column queries."""
    queries = list(queries)
    Build(api, range(table_size), queries, queries)


def Build(api, table_values, queries, row_index):
    """table_values: the constant table entries t_0 .. t_(n-1); queries: the looked-up values;
    row_index[i]: a variable holding the table row query i claims (only the multiplicity HINT uses
    it: a wrong index makes the argument fail, it cannot make a wrong query pass)."""
    queries, row_index = list(queries), list(row_index)
    table_values = [int(t) for t in table_values]
    if not queries:
        return
    mults = api.NewHintCount(row_index, len(table_values))

    def cb(api, challenge):
        # same constraints as gnark's DivUnchecked / Inverse per row; the witness side shares one
        # inversion per argument (api.DivUncheckedBatch / InverseBatch)
        lp = api.Sum(api.DivUncheckedBatch(mults, [api.Sub(challenge, t) for t in table_values]))
        rp = api.Sum(api.InverseBatch([api.Sub(challenge, q) for q in queries]))
        api.AssertIsEqual(lp, rp)

    multicommit.WithCommitment(api, cb, *queries, *mults)
