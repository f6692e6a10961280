"""Share of a host-mirror file's code lines that also occur, verbatim, in the reference file it cites (VERDICT r2 item 7:
below 15 % per file).  Reported twice: all lines, and without the def / class / import lines whose text the mirrored API fixes.  Lines are stripped, docstrings and comments dropped, lines of 12 characters or fewer ignored.
Lines are LOGICAL lines with the layout removed (logical_lines below; --physical: the physical lines of round 3).
Needs /root/reference (this container only).   python tools/shared_lines.py [-v] [--physical]"""
import ast
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pygrank"
PAIRS = [("pygrank_amd/signals.py", "core/signals.py"), ("pygrank_amd/convergence.py", "algorithms/convergence.py"),
         ("pygrank_amd/postprocess.py", "algorithms/postprocess/postprocess.py"), ("pygrank_amd/filters.py", "algorithms/filters/abstract_filters.py"),
         ("pygrank_amd/filters.py", "algorithms/filters/adhoc.py"), ("pygrank_amd/measures.py", "measures/unsupervised.py"),
         ("pygrank_amd/measures.py", "measures/supervised.py"), ("pygrank_amd/backend/__init__.py", "core/backend/__init__.py"),
         ("pygrank_amd/utils.py", "core/utils.py"), ("pygrank_amd/preprocessing.py", "core/utils/preprocessing.py"),
         ("pygrank_amd/preprocessing.py", "fastgraph/wrapgraph.py"), ("pygrank_amd/backend/hip.py", "core/backend/numpy.py"),
         ("pygrank_amd/backend/hip.py", "core/backend/pytorch.py")]


def code_lines(path):
    src = open(path).read()
    drop = set()
    try:
        for node in ast.walk(ast.parse(src)):
            if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef)) and node.body and isinstance(node.body[0], ast.Expr) \
                    and isinstance(getattr(node.body[0], "value", None), ast.Constant) and isinstance(node.body[0].value.value, str):
                drop.update(range(node.body[0].lineno, node.body[0].end_lineno + 1))
    except SyntaxError:
        pass
    comments = {}
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.COMMENT:
            comments[tok.start[0]] = tok.start[1]
    out = []
    for number, line in enumerate(src.splitlines(), 1):
        if number in drop:
            continue
        if number in comments:
            line = line[:comments[number]]
        line = line.strip()
        if len(line) > 12:
            out.append(line)
    return out


def logical_lines(path):
    """The file's LOGICAL lines (a statement wrapped over several physical lines is one), comments and docstrings dropped, all
    layout removed (tokens joined, a blank only between two words): re-wrapping a copied statement does not hide it
    (Supervised.to_numpy was the reference's text, re-wrapped)."""
    src = open(path).read()
    out, words, depth_doc = [], [], True
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER):
            continue
        if tok.type == tokenize.NEWLINE:
            only_string = len(words) == 1 and words[0][0] == tokenize.STRING
            text = ""
            for kind, string in words:
                if text and kind in (tokenize.NAME, tokenize.NUMBER) and (text[-1].isalnum() or text[-1] == "_"):
                    text += " "
                text += string
            if not only_string and len(text) > 12:
                out.append(text)
            words = []
            continue
        words.append((tok.type, tok.string))
    return out


def main():
    verbose = "-v" in sys.argv
    lines_of = logical_lines if "--physical" not in sys.argv else code_lines
    totals = {}
    for mine, theirs in PAIRS:
        ref_path = os.path.normpath(os.path.join(REF, theirs))
        if not os.path.exists(ref_path):
            continue
        ref = set(lines_of(ref_path))
        lines = lines_of(os.path.join(ROOT, mine))
        hits = totals.setdefault(mine, (set(), len(lines)))[0]
        for index, line in enumerate(lines):
            if line in ref:
                hits.add((index, line))
    for mine, (hits, count) in totals.items():
        forced = sum(1 for _, line in hits if line.startswith(("def ", "class ", "import ", "from ", "@")))
        print(f"{mine}: {len(hits)} / {count} = {100.0 * len(hits) / max(count, 1):.0f} %   (signatures / imports the API fixes: {forced}; "
              f"other lines: {len(hits) - forced} = {100.0 * (len(hits) - forced) / max(count, 1):.0f} %)")
        if verbose:
            for _, line in sorted(hits):
                print("      " + line)


if __name__ == "__main__":
    main()
