"""Development check: every `extern "C"` launcher declared in csrc/launchers.h has the same parameter TYPE list as its
definition in the .hip files (extern "C" symbols carry no signature, so a drifted declaration links fine and corrupts
the call)."""
import glob
import os
import re
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "poem-v2_amd", "csrc")


def types(params):
    out = []
    for a in params.split(","):
        a = re.sub(r"\b\w+$", "", a.strip()).strip()
        out.append(a.replace(" ", ""))
    return out


def main():
    decl = open(os.path.join(CSRC, "launchers.h")).read()
    decls = {m.group(2): re.sub(r"\s+", " ", m.group(3))
             for m in re.finditer(r"(hipError_t|size_t)\s+(poem_\w+)\s*\(([^;]*?)\)\s*;", decl, re.S)}
    defs = {}
    for f in glob.glob(os.path.join(CSRC, "*.hip")):
        for m in re.finditer(r'extern "C"\s+(hipError_t|size_t)\s+(poem_\w+)\s*\(([^{;]*?)\)\s*\{', open(f).read(), re.S):
            defs[m.group(2)] = re.sub(r"\s+", " ", m.group(3))
    bad = [k for k, v in decls.items() if k not in defs or types(v) != types(defs[k])]
    return decls, bad


if __name__ == "__main__":
    d, bad = main()
    print(f"{len(d)} declarations checked, mismatches: {bad}")
    sys.exit(1 if bad else 0)
