"""Truth-table export: the files the reference writes for one ``Block_TT`` filter.

Counterpart of ``Block_TT.get_TT_block_1filter`` / ``for_1_filter`` / ``save_cnf_dnf`` /
``get_expresion_methode1`` and ``get_exp_with_y`` (models/TT_FHE_SMALL.py:251-275, :344-431), fed
from the truth tables the plan built on the GPU (``model.get_table``, canonical order: pattern =
index read MSB first over (c, kh, kw), TT_FHE_SMALL.py:330-334) instead of a float forward over
all 2^n patterns.  For every non-constant filter:

    Truth_Table_block{B}_filter_{f}_coefdefault_{v}_sousblock_{S}.csv   index, the n input bits, the filter's column
    DNF_expression_block{B}_filter_{f}_coefdefault_{v}_sousblock_{S}.txt   minimal sum of products (sympy SOPform)
    CNF_expression_block{B}_filter_{f}_coefdefault_{v}_sousblock_{S}.txt   minimal product of sums (sympy POSform)
    table_outputblock_{B}_filter_{f}_coefdefault_{v}.txt                    CNF of (y <-> filter), the SAT-solver form

Expressions are produced for n <= ``max_expr_bits`` inputs (the reference: n in {4, 8, 9} only);
for the 16-input tables of TT-small only the CSV is practical.  Unlike the reference's exporter,
grouped blocks (several input channels per group) work too: the table of group g is used for
its filters.  Host-side Python (pandas + sympy), not a hot path.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import numpy as np


def pattern_frame(n: int):
    """Index column + the n input bits of every pattern, MSB first (TT_FHE_SMALL.py:330-332)."""
    import pandas as pd
    idx = np.arange(2 ** n, dtype=np.int64)
    bits = ((idx[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1).astype(np.int64)
    return pd.DataFrame(bits).reset_index()


def minimal_forms(minterms: List[int], n: int):
    """(DNF, CNF) of the function that is 1 exactly on ``minterms`` (TT_FHE_SMALL.py:405-427)."""
    from sympy import symbols
    from sympy.logic import POSform, SOPform
    xs = symbols(", ".join(f"x_{i}" for i in range(n)))
    xs = list(xs) if n > 1 else [xs]
    return SOPform(xs, minterms=minterms), POSform(xs, minterms=minterms)


def cnf_with_output(dnf, cnf) -> str:
    """CNF of ``y <-> f`` in the reference's text format (TT_FHE_SMALL.py:251-275): one clause
    ``(y | ~l1 | ~l2 ...)`` per DNF term (term -> y) and ``(clause | ~y)`` per CNF clause."""
    def lits(text: str, sep: str) -> List[str]:
        return [t for t in text.replace("(", "").replace(")", "").split(sep) if t]

    dnf_s, cnf_s = str(dnf).replace(" ", ""), str(cnf).replace(" ", "")
    clauses = []
    for term in dnf_s.split("|"):
        neg = [l[1:] if l.startswith("~") else "~" + l for l in lits(term, "&")]
        clauses.append("(y | " + " | ".join(neg) + ")")
    for clause in cnf_s.split("&"):
        clauses.append("(" + " | ".join(lits(clause, "|")) + " | ~y)")
    return " & ".join(clauses)


def export_filter(column: np.ndarray, n: int, filter_index: int, out_dir: str, block: int, sub_block: int,
                  max_expr_bits: int = 9) -> Dict[str, Optional[str]]:
    """Files for one filter.  ``column``: its 2^n table entries (0/1), canonical order."""
    import pandas as pd
    os.makedirs(out_dir, exist_ok=True)
    col = np.asarray(column).astype(np.float32)
    uniq = np.unique(col)
    prefix = os.path.join(out_dir, "")
    out: Dict[str, Optional[str]] = {"dnf": None, "cnf": None, "cnf_with_y": None, "csv": None}
    if len(uniq) == 1:                                   # constant filter: the value only (:351-361)
        with open(f"{prefix}table_outputblock_{block}_filter_{filter_index}_coefdefault_{uniq[0]}.txt", "w") as f:
            f.write(str(uniq[0]))
        out["cnf_with_y"] = str(uniq[0])
        return out
    for value in uniq[1:]:
        answer = col == value
        frame = pd.concat([pattern_frame(n), pd.DataFrame(answer, columns=[f"Filter_{filter_index}_Value_{int(value)}"])], axis=1)
        csv = f"{prefix}Truth_Table_block{block}_filter_{filter_index}_coefdefault_{value}_sousblock_{sub_block}.csv"
        frame.to_csv(csv)
        out["csv"] = csv
        if n <= max_expr_bits:
            dnf, cnf = minimal_forms(frame["index"].values[answer].tolist(), n)
            y = cnf_with_output(dnf, cnf)
            out.update(dnf=str(dnf), cnf=str(cnf), cnf_with_y=y)
            with open(f"{prefix}table_outputblock_{block}_filter_{filter_index}_coefdefault_{value}.txt", "w") as f:
                f.write(y)
            with open(f"{prefix}CNF_expression_block{block}_filter_{filter_index}_coefdefault_{value}_sousblock_{sub_block}.txt", "w") as f:
                f.write(str(cnf))
            with open(f"{prefix}DNF_expression_block{block}_filter_{filter_index}_coefdefault_{value}_sousblock_{sub_block}.txt", "w") as f:
                f.write(str(dnf))
    return out


def export_block(table: np.ndarray, out_dir: str, block: int, sub_block: int, filters: Optional[Iterable[int]] = None,
                 max_expr_bits: int = 9) -> Dict[int, Dict[str, Optional[str]]]:
    """``table``: [groups][2^n][cout_g] bits as returned by ``model.get_table(name)`` (or by the
    oracle's ``build_lut``).  Filter f = output channel f of the block = (group f // cout_g,
    output f % cout_g)."""
    g, size, cout_g = table.shape
    n = int(size).bit_length() - 1
    assert 2 ** n == size
    todo = range(g * cout_g) if filters is None else filters
    return {f: export_filter(table[f // cout_g, :, f % cout_g], n, f, out_dir, block, sub_block, max_expr_bits) for f in todo}


def literal_count(expr_text: Optional[str]) -> int:
    """Number of literals of an expression string (a gate-count proxy: one input per literal)."""
    return 0 if not expr_text else expr_text.count("x_")
