"""Self-check against transliteration: for every fiat_amd/*.py with a same-named file in the reference
(build container only), the share of this file's code tokens (comments and docstrings removed) that appear,
in order, in the reference file (longest common subsequence / own length).  Independent restatements of a
shared API land around 0.3-0.5 (fiat_amd/dual_set.py: 0.34); a condensed copy scores > 0.75."""
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/FIAT"


def code_tokens(path):
    out, prev = [], None
    with open(path, "rb") as f:
        for tok in tokenize.tokenize(f.readline):
            if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT,
                            tokenize.ENCODING, tokenize.ENDMARKER):
                continue
            if tok.type == tokenize.STRING and prev in (None, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.NL):
                prev = tok.type
                continue   # docstring
            prev = tok.type
            if tok.type != tokenize.OP:     # names, keywords, literals: punctuation says nothing about origin
                out.append(tok.string)
    return out


def lcs(a, b):
    row = [0] * (len(b) + 1)
    for x in a:
        diag = 0
        for j, y in enumerate(b, 1):
            diag, row[j] = row[j], (diag + 1 if x == y else max(row[j], row[j - 1]))
    return row[-1]


if __name__ == "__main__":
    rows = []
    for name in sorted(os.listdir(os.path.join(ROOT, "fiat_amd"))):
        ref = os.path.join(REF, name)
        if name.endswith(".py") and os.path.exists(ref):
            mine = code_tokens(os.path.join(ROOT, "fiat_amd", name))
            theirs = code_tokens(ref)
            rows.append((lcs(mine, theirs) / max(1, len(mine)), name, len(mine), len(theirs)))
    for frac, name, n, m in sorted(rows, reverse=True):
        print(f"{frac:5.2f}  {name:36s} {n:6d} tokens (reference {m})")
