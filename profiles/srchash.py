#!/usr/bin/env python3
"""Ties a committed rocprofv3 counter summary to the kernel source it was measured on.

`kernel_source_hashes(root)` maps every `__global__` kernel of halo2_vectordb_amd/csrc/*.hip to the SHA-256 of the source it is
compiled from: its .hip file and every local header that file includes, transitively — the CODE of those files: comments are
dropped and runs of white space count as one blank, so a header whose documentation was edited does not disown a profile, while any
token that reaches the compiler does.  tools/profile_round.sh records the map on
the GPU box beside the counter passes (the snapshot there IS the profiled source), profiles/summarize.py writes it into the
header of profiles/<tag>_pmc_summary.csv, and bench.py reports `roofline.traffic` from a summary only when the dominant kernel's
hash in it equals the hash of the tree the bench runs from — otherwise `traffic: null, traffic_stale: true`.
usage: python profiles/srchash.py [root]   -> JSON map on stdout"""
import hashlib
import json
import os
import re
import sys

_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
# string and character literals are kept as they are (a "//" inside one is not a comment); comments become a blank
_LEX = re.compile(r'"(?:\\.|[^"\\\n])*"|\'(?:\\.|[^\'\\\n])*\'|//[^\n]*|/\*.*?\*/', re.S)


def code_of(text):
    """the text as the compiler's front end sees it, near enough: no comments, white space collapsed"""
    t = _LEX.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return re.sub(r"\s+", " ", t).strip()


_KERNEL = re.compile(r'__global__\s+(?:__launch_bounds__\s*\([^)]*\)\s*)?(?:static\s+)?void\s+([A-Za-z_]\w*)\s*\(')


def _closure(path, dirs, seen):
    path = os.path.realpath(path)
    if path in seen:
        return
    seen.add(path)
    text = open(path, errors="replace").read()
    for inc in _INC.findall(text):
        for d in [os.path.dirname(path)] + dirs:
            cand = os.path.join(d, inc)
            if os.path.isfile(cand):
                _closure(cand, dirs, seen)
                break


def kernel_source_hashes(root):
    csrc = os.path.join(root, "halo2_vectordb_amd", "csrc")
    dirs = [csrc, os.path.join(root, "include")]
    out = {}
    for name in sorted(os.listdir(csrc)):
        if not name.endswith(".hip"):
            continue
        path = os.path.join(csrc, name)
        seen = set()
        _closure(path, dirs, seen)
        h = hashlib.sha256()
        for f in sorted(seen):
            h.update(os.path.basename(f).encode() + b"\0")
            h.update(code_of(open(f, errors="replace").read()).encode())
        digest = h.hexdigest()
        for kernel in set(_KERNEL.findall(open(path, errors="replace").read())):
            out[kernel] = digest
    return out


if __name__ == "__main__":
    print(json.dumps(kernel_source_hashes(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), sort_keys=True))
