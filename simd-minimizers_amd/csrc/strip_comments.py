#!/usr/bin/env python3
"""Strips C / C++ comments from a header before it is embedded as the source text of the run-time specialisation
(mm_jit.hip): the shipped library then carries the kernel source without the lab notes - and without the names of the
experiment switches those notes mention.  String and character literals are kept as they are; line structure is kept
(one output line per input line) so that hiprtc's diagnostics still point at the right line of the header."""
import sys


def strip(src: str) -> str:
    out = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            # a comment that ends in a backslash continues a macro line: keep the continuation
            if j > i and src[j - 1] == "\\":
                out.append("\\")
            i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("\n" * src.count("\n", i, j) or " ")
            i = j
        else:
            out.append(c)
            i += 1
    return "\n".join(line.rstrip() for line in "".join(out).split("\n"))


if __name__ == "__main__":
    sys.stdout.write(strip(open(sys.argv[1]).read()))
