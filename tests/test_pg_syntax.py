"""The PostgreSQL branches of this repo's own sources (-DCRYO_HAVE_POSTGRES: the CryoRelOps adapter pg/cryo_pg_rel.c and
the GUC registration / elog paths of pg_cryogen_amd/host/*.c) cannot be built in an image without a PostgreSQL server.
This test only makes sure they parse and type-check: gcc -fsyntax-only against the declaration-only stand-ins of
tests/pg_stubs/ (names and arities of the PostgreSQL 12/13 functions they call; nothing is linked or run)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = [os.path.join(ROOT, "pg", "cryo_pg_rel.c")] + sorted(glob.glob(os.path.join(ROOT, "pg_cryogen_amd", "host", "*.c")))


@pytest.mark.parametrize("src", SOURCES, ids=[os.path.relpath(s, ROOT) for s in SOURCES])
def test_postgres_branch_parses(src):
    cmd = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-DCRYO_HAVE_POSTGRES",
           "-I" + os.path.join(ROOT, "tests", "pg_stubs"), "-I" + os.path.join(ROOT, "pg_cryogen_amd", "host"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pg"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_adapter_implements_every_rel_op():
    """the ops table of the adapter has one entry per member of CryoRelOps, in order"""
    hdr = open(os.path.join(ROOT, "pg_cryogen_amd", "host", "staging.h")).read()
    body = hdr[hdr.index("typedef struct CryoRelOps"):]
    body = body[:body.index("} CryoRelOps")]
    members = [ln.split("(*")[1].split(")")[0] for ln in body.splitlines() if "(*" in ln]
    src = open(os.path.join(ROOT, "pg", "cryo_pg_rel.c")).read()
    tab = src[src.index("static const CryoRelOps cryo_pg_ops = {"):]
    tab = tab[tab.index("{") + 1:tab.index("};")]
    entries = [e.strip() for e in tab.replace("\n", " ").split(",") if e.strip()]
    assert len(members) >= 7 and entries == ["pg_" + m for m in members], (members, entries)
