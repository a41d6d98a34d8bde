#!/usr/bin/env python3
"""Keeps the measured figures of DESIGN.md identical to the files under profiles/ they come from.

Two mechanisms, both checked by tests/test_design_citations.py (-m "not gpu"):

1. Extract blocks.  Between
       <!-- extract: r03_blocks.txt grep=^chain -->
       <!-- /extract -->
   this script writes the lines of profiles/r03_blocks.txt that match the regular expression (all lines without
   `grep=`; `cut=N` limits the line length) as a fenced code block: the table IS the file's text.
2. Inline citations.  A figure in the prose is followed by the file it was read from in square brackets,
       150926 ns [r03_bench_kernel_stats.csv]
   and must occur literally in that file.

usage: scripts/design_extracts.py [--check]   (rewrites DESIGN.md, or fails if it would change)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DESIGN = os.path.join(ROOT, "DESIGN.md")
PROFILES = os.path.join(ROOT, "profiles")
BLOCK = re.compile(r"(<!-- extract: (\S+)((?: \w+=\S+)*) -->\n)(.*?)(<!-- /extract -->)", re.S)
CITE = re.compile(r"(?<![\w.])(\d[\d ,.]*\d|\d)\s*(?:ms|us|µs|ns|%|x|W|MHz|GHz|MB|KB|GB/s|TB/s|GSamples/s|MSamples/s|launches|bytes)?\s*\[((?:r\d\d[a-z]?_|traffic_)[\w.]+)\]")


def render(fname, opts):
    path = os.path.join(PROFILES, fname)
    lines = open(path, errors="replace").read().splitlines()
    o = dict(kv.split("=", 1) for kv in opts.split())
    if "grep" in o:
        rx = re.compile(o["grep"])
        lines = [l for l in lines if rx.search(l)]
    if "cut" in o:
        lines = [l[: int(o["cut"])] for l in lines]
    return "```\n" + "\n".join(l.rstrip() for l in lines) + "\n```\n"


def rewrite(text):
    return BLOCK.sub(lambda m: m.group(1) + render(m.group(2), m.group(3)) + m.group(5), text)


def citations(text):
    """(figure, file) pairs quoted in the prose (outside the extract blocks)."""
    prose = BLOCK.sub("", text)
    return [(m.group(1).strip(), m.group(2)) for m in CITE.finditer(prose)]


def main():
    text = open(DESIGN).read()
    new = rewrite(text)
    if "--check" in sys.argv:
        if new != text:
            sys.exit("DESIGN.md's extract blocks differ from profiles/: run scripts/design_extracts.py")
        bad = []
        for fig, fname in citations(text):
            path = os.path.join(PROFILES, fname)
            if not os.path.exists(path) or fig.replace(" ", "") not in open(path, errors="replace").read().replace(" ", ""):
                bad.append((fig, fname))
        if bad:
            sys.exit(f"figures not found in the files they cite: {bad}")
        print(f"ok: {len(BLOCK.findall(text))} extract blocks, {len(citations(text))} inline citations")
        return
    open(DESIGN, "w").write(new)
    print("DESIGN.md rewritten")


if __name__ == "__main__":
    main()
