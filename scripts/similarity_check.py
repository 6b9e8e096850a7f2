"""Development aid (reads /root/reference, so it only runs in the build container): how many of a file's code lines also
occur in given reference files.  Two figures, both over normalised lines (comments / blank lines / docstrings dropped,
whitespace collapsed): `set` = share of this file's lines that occur anywhere in the reference files, `seq` = share covered
by difflib's matching blocks against the concatenation.

    python scripts/similarity_check.py seamless_communication_amd/streaming/agents.py /root/reference/src/seamless_communication/streaming/agents/*.py
"""
import ast
import difflib
import io
import re
import sys
import tokenize


def code_lines(path):
    src = open(path).read()
    doc_lines = set()
    try:
        tree = ast.parse(src)
        for node in ast.walk(tree):
            if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef, ast.AsyncFunctionDef)) and node.body:
                first = node.body[0]
                if isinstance(first, ast.Expr) and isinstance(getattr(first, "value", None), ast.Constant) and isinstance(first.value.value, str):
                    doc_lines.update(range(first.lineno, first.end_lineno + 1))
    except SyntaxError:
        pass
    comment_cols = {}
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.COMMENT:
            comment_cols[tok.start[0]] = tok.start[1]
    out = []
    for no, line in enumerate(src.splitlines(), 1):
        if no in doc_lines:
            continue
        if no in comment_cols:
            line = line[: comment_cols[no]]
        line = re.sub(r"\s+", " ", line).strip().rstrip(",")
        if line and line not in (")", "(", "]", "[", "}", "{", "):", "else:", "try:", "pass"):
            out.append(line)
    return out


def main():
    verbose = "-v" in sys.argv
    files = [a for a in sys.argv[1:] if a != "-v"]
    mine = code_lines(files[0])
    ref = [l for p in files[1:] for l in code_lines(p)]
    ref_set = set(ref)
    hit = sum(1 for l in mine if l in ref_set)
    sm = difflib.SequenceMatcher(None, mine, ref, autojunk=False)
    seq = sum(b.size for b in sm.get_matching_blocks())
    print(f"{files[0]}: {len(mine)} code lines; set {hit} ({100 * hit / len(mine):.1f} %), seq {seq} ({100 * seq / len(mine):.1f} %)")
    if verbose:
        for l in mine:
            if l in ref_set:
                print("   ", l)


if __name__ == "__main__":
    main()
