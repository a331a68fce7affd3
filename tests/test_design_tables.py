"""DESIGN.md section 6's table is generated from the committed rocprofv3 summaries (tools/gen_design_tables.py): the
prose cannot drift from the profiles it cites (round 2's had).  CPU only."""
import importlib.util
import os

from tests.oracle_lib import ROOT

spec = importlib.util.spec_from_file_location("gen_design_tables", os.path.join(ROOT, "tools", "gen_design_tables.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def test_design_table_matches_the_profiles():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, z = design.index(gen.BEGIN), design.index(gen.END) + len(gen.END)
    assert design[a:z] == gen.block(), "DESIGN.md is stale: run `python tools/gen_design_tables.py --write`"


def test_every_row_comes_from_a_round_3_summary():
    rows = [l for l in gen.block().splitlines() if l.startswith("| ") and "`r0" in l]
    assert len(rows) >= 15
    assert all("`r03_" in l for l in rows), [l for l in rows if "`r03_" not in l]
    # the headline's numbers are the ones the verdict recomputed by hand: ~2.2-2.3 GHz held, ~4.2 cycles per VALU instruction
    head = next(l for l in rows if "(headline)" in l).split("|")
    assert 2.1 < float(head[4]) < 2.35 and 4.0 < float(head[6]) < 4.4
