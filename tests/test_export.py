"""truth-table export (CSV / DNF / CNF / SAT form) against the files the
reference's own exporter wrote for the same table (tests/golden/ref_export_xsmall.json, made by
oracle/gen_golden.py from Block_TT.get_TT_block_1filter, models/TT_FHE_SMALL.py:344-431)."""
import json
import os

import numpy as np
import pytest

from scale_imagenet_amd import export as E
from _util import GOLD as GOLDEN_DIR, spec_and_state


def golden():
    with open(os.path.join(GOLDEN_DIR, "ref_export_xsmall.json")) as f:
        return json.load(f)


def check_against_golden(table: np.ndarray, tmp_path) -> int:
    g = golden()
    compared = 0
    for f_str, want in g["filters"].items():
        f = int(f_str)
        if table[f, :, 0].astype(int).tolist() != want["column"]:
            continue                                  # a near-tie entry decided the other way: another function
        got = E.export_block(table, str(tmp_path / f_str), g["blockici"], g["sousblockici"], filters=[f])[f]
        assert got["dnf"] == want["dnf"] and got["cnf"] == want["cnf"], f
        assert got["cnf_with_y"] == want["cnf_with_y"] or want["cnf_with_y"] is None, f
        files = {n: open(tmp_path / f_str / n).read() for n in sorted(os.listdir(tmp_path / f_str))}
        assert files == want["files"], (f, sorted(files), sorted(want["files"]))
        compared += 1
    return compared


def test_export_matches_reference_files(tmp_path):
    from oracle import ttnet_bits as OB
    spec, st = spec_and_state("xsmall")
    table, _ = OB.build_lut(st, spec.blocks[0].conv1)
    assert check_against_golden(table, tmp_path) >= 10


def test_cnf_with_output_is_equivalence():
    """The SAT form encodes y <-> f: checked by brute force on a 4-input function."""
    from sympy import symbols
    from sympy.logic.boolalg import to_cnf
    from sympy.parsing.sympy_parser import parse_expr
    minterms = [1, 2, 7, 8, 13]
    dnf, cnf = E.minimal_forms(minterms, 4)
    text = E.cnf_with_output(dnf, cnf)
    names = {f"x_{i}": symbols(f"x_{i}") for i in range(4)}
    names["y"] = symbols("y")
    expr = parse_expr(text, local_dict=names)
    for idx in range(16):
        bits = {names[f"x_{i}"]: bool((idx >> (3 - i)) & 1) for i in range(4)}      # x_0 = MSB
        f_val = idx in minterms
        assert bool(expr.subs({**bits, names["y"]: f_val})) is True
        assert bool(expr.subs({**bits, names["y"]: not f_val})) is False


def test_sixteen_input_block_exports_csv_only(tmp_path):
    rng = np.random.default_rng(0)
    table = rng.integers(0, 2, size=(1, 65536, 2)).astype(np.uint8)
    out = E.export_block(table, str(tmp_path), 4, 0, filters=[1])[1]
    assert out["dnf"] is None and out["csv"] is not None
    lines = open(out["csv"]).read().splitlines()
    assert len(lines) == 65537 and lines[0].split(",")[1] == "index" and lines[0].split(",")[-1] == "Filter_1_Value_1"
    assert lines[1 + 5].split(",")[2:18] == list("0000000000000101")       # pattern 5, MSB first
