#!/usr/bin/env python
"""Keeps a Markdown file readable in a terminal / diff view: no line longer than WIDTH characters (housekeeping).
  * a table with a row longer than WIDTH becomes a bullet list -- `first cell` -- other cells --, wrapped, its header kept as an italic caption;
  * a paragraph / bullet line longer than WIDTH is wrapped (continuation lines indented under the bullet's text);
  * fenced code blocks and tables that fit are left as they are.
usage: python tools/wrap_md.py FILE [--check]      (--check: exit 1 if a line outside code fences exceeds WIDTH)"""
import re, sys, textwrap
WIDTH = 160


def wrap_line(line):
    m = re.match(r"^(\s*(?:[-*]|\d+\.)\s+)", line)
    indent = " " * len(m.group(1)) if m else re.match(r"^\s*", line).group(0)
    first = m.group(1) if m else indent
    body = line[len(first):]
    out = textwrap.wrap(body, WIDTH - len(indent), break_long_words=False, break_on_hyphens=False)
    return [first + out[0]] + [indent + o for o in out[1:]] if out else [line]


def cells(row):
    return [c.strip() for c in re.split(r"(?<!\\)\|", row.strip().strip("|"))]


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    if "--check" in sys.argv:
        fence, bad = False, 0
        for i, l in enumerate(lines, 1):
            if l.startswith("```"):
                fence = not fence
            elif not fence and len(l) > WIDTH:
                print(f"{path}:{i}: {len(l)} characters"); bad += 1
        sys.exit(1 if bad else 0)
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.startswith("```"):
            fence = not fence
            out.append(l); i += 1; continue
        if fence:
            out.append(l); i += 1; continue
        if l.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            tab = lines[i:j]
            if max(len(t) for t in tab) <= WIDTH:
                out += tab
            else:
                head = cells(tab[0])
                out += wrap_line("*" + " — ".join(head) + ":*")
                out.append("")
                for row in tab[2:]:
                    cs = cells(row)
                    out += wrap_line("- " + cs[0] + " — " + " — ".join(c for c in cs[1:] if c))
            i = j
            continue
        out += wrap_line(l) if len(l) > WIDTH else [l]
        i += 1
    open(path, "w").write("\n".join(out))


main()
