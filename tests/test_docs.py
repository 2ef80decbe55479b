"""CPU tier: DESIGN.md describes the code that exists.  Every `` `symbol` (`file`) `` pair of DESIGN.md sections 1 and 4 must grep
true: the file exists and contains the symbol as a word.  (Round 3's document still cited kernels that had been replaced.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIR = re.compile(r"`([A-Za-z_][A-Za-z0-9_]*)`((?:, `[A-Za-z_][A-Za-z0-9_]*`)*) \(`((?:csrc|include|tools|tests|oracle)/[A-Za-z0-9_./-]+)`\)")


def _section(text, number):
    m = re.search(r"^## %d\. .*?(?=^## \d+\. )" % number, text, re.S | re.M)
    assert m, "DESIGN.md has no section %d" % number
    return m.group(0)


def _resolve(path):
    p = os.path.join(ROOT, "crystals-kyber_amd", path) if path.startswith("csrc/") else os.path.join(ROOT, path)
    return p


def test_design_cites_symbols_that_exist():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(text.splitlines()) <= 300, "DESIGN.md is the current-state document: history belongs in LABNOTES.md"
    pairs = []
    for sec in (1, 4):
        for m in PAIR.finditer(_section(text, sec)):
            syms = [m.group(1)] + re.findall(r"`([A-Za-z_][A-Za-z0-9_]*)`", m.group(2))
            pairs += [(s, m.group(3)) for s in syms]
    assert len(pairs) >= 60, "citation convention `symbol` (`file`) not found often enough: %d" % len(pairs)
    missing = []
    cache = {}
    for sym, path in pairs:
        p = _resolve(path)
        if p not in cache:
            cache[p] = open(p).read() if os.path.exists(p) else None
        if cache[p] is None or not re.search(r"\b%s\b" % re.escape(sym), cache[p]):
            missing.append((sym, path))
    assert not missing, missing


def test_documents_point_at_files_that_exist():
    """every profiles/..., tools/..., tests/... path quoted in backticks in DESIGN.md / README.md / INTEGRATION.md exists
    (generated files of the CURRENT round may be absent in a fresh clone only if the round has not been profiled yet)"""
    bad = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(re.findall(r"`((?:profiles|tools|tests|oracle|include)/[A-Za-z0-9_./-]+\.[a-z0-9]+)`", text)):
            if "*" in path or "rNN" in path or "<" in path:
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                bad.append((doc, path))
    assert not bad, bad
