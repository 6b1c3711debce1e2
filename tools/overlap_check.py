#!/usr/bin/env python
"""How much of each mdapy_amd/*.py is found verbatim in the same-named reference file: the fraction of a file's tokens
(comments and docstrings stripped) that lie in a run of >= 8 consecutive tokens which also occurs in the reference file.
Build-container tool (reads /root/reference); usage: python tools/overlap_check.py [threshold]"""
import io, os, sys, tokenize

REF = "/root/reference/src/mdapy"
OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mdapy_amd")
RUN = 8


def toks(path):
    out = []
    src = open(path, encoding="utf-8").read()
    prev_sig = None
    for t in tokenize.generate_tokens(io.StringIO(src).readline):
        if t.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING, tokenize.ENDMARKER):
            if t.type in (tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT):
                prev_sig = "stmt"
            continue
        if t.type == tokenize.STRING and prev_sig in (None, "stmt"):  # an expression statement that is a string: docstring
            prev_sig = "doc"
            continue
        out.append(t.string)
        prev_sig = "tok"
    return out


def overlap(a, b):
    grams = set(tuple(b[i:i + RUN]) for i in range(len(b) - RUN + 1))
    hit = [False] * len(a)
    for i in range(len(a) - RUN + 1):
        if tuple(a[i:i + RUN]) in grams:
            for k in range(i, i + RUN):
                hit[k] = True
    return sum(hit) / max(len(a), 1)


def main():
    thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.35
    rows = []
    for f in sorted(os.listdir(OURS)):
        if not f.endswith(".py"):
            continue
        r = os.path.join(REF, f)
        if not os.path.exists(r):
            continue
        a, b = toks(os.path.join(OURS, f)), toks(r)
        rows.append((overlap(a, b), f, len(a)))
    bad = 0
    for o, f, n in sorted(rows, reverse=True):
        flag = "  <-- over" if o >= thr else ""
        bad += o >= thr
        print(f"{o:5.2f}  {f:40s} {n:6d} tokens{flag}")
    print(f"{bad} file(s) at or above {thr}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
