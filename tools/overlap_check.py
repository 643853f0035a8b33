#!/usr/bin/env python
"""tools/overlap_check.py [files...] -- line overlap of product files with the reference checkout (build container only): whitespace- and
quote-normalised code lines longer than 25 characters that also occur in any file under $REF/lib.  A guard against transcribing the
reference's host layer instead of designing it (the kernels and their hosts must be original; API surface lines -- signatures, config keys --
inevitably coincide)."""
import glob
import os
import re
import sys

REF = os.environ.get("REF", "/root/reference")


def norm(line):
    line = line.split("#")[0].strip().replace('"', "'")
    return re.sub(r"\s+", "", line)


def lines_of(path):
    return [n for n in (norm(l) for l in open(path, errors="ignore")) if len(n) > 25]


def main():
    ref = set()
    for f in glob.glob(os.path.join(REF, "lib", "**", "*.py"), recursive=True):
        ref.update(lines_of(f))
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "ssdnerf_amd", "**", "*.py"), recursive=True))
    for f in files:
        own = lines_of(f)
        shared = [l for l in own if l in ref]
        api = [l for l in shared if l.startswith(("def", "class", "@")) or ".get('" in l or l.endswith(("=None,", "=None):"))]
        print(f"{os.path.relpath(f):42s} code lines {len(own):4d}  shared {len(shared):3d} ({100 * len(shared) / max(len(own), 1):4.1f} %)  "
              f"of which signatures / cfg keys {len(api):3d}  -> {100 * (len(shared) - len(api)) / max(len(own), 1):4.1f} % other")
        if os.environ.get("VERBOSE"):
            for l in shared:
                if l not in api:
                    print("      ", l[:150])


if __name__ == "__main__":
    main()
