"""DESIGN.md's measured figures against the files under profiles/ they cite: the extract blocks must be
exactly what scripts/design_extracts.py renders from the cited file, and every figure quoted in the prose with a
`[file]` behind it must occur literally in that file."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_design_extract_blocks_and_inline_citations_match_the_profiles():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_extracts.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_design_quotes_enough_to_be_worth_checking():
    import design_extracts as de

    text = open(de.DESIGN).read()
    cites = de.citations(text)
    # (the headline figures are extract blocks of profiles/r03_summary.txt and r03a_summary.txt; the prose carries the rest)
    assert len(de.BLOCK.findall(text)) >= 12 and len(cites) >= 6
    # every cited file exists and belongs to the round's set (or is the traffic file the bench line reads)
    for _fig, fname in cites:
        assert os.path.exists(os.path.join(de.PROFILES, fname)), fname
    # no profiles/rNN_* path is named in the text without existing
    import re

    for m in re.finditer(r"profiles/((?:r\d\d[a-z]?_|traffic_)[\w.]+?\.(?:txt|csv|json))", text):
        assert os.path.exists(os.path.join(de.PROFILES, m.group(1))), m.group(1)
