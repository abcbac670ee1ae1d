#!/usr/bin/env python3
"""Reflow markdown prose to <= 120 columns: paragraphs and list items are re-wrapped, tables, code fences, headings, block quotes
and lines that are one unbreakable token stay as they are."""
import re, sys, textwrap
W = 120
def flush(buf, out):
    if not buf:
        return
    first = buf[0]
    m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", first)
    if m:
        indent = m.group(1) + " " * (len(m.group(0)) - len(m.group(1)))
        head = m.group(0)
        text = " ".join([first[len(head):].strip()] + [b.strip() for b in buf[1:]])
        out.extend(textwrap.wrap(text, W, initial_indent=head, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    else:
        indent = re.match(r"^\s*", first).group(0)
        text = " ".join(b.strip() for b in buf)
        out.extend(textwrap.wrap(text, W, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    buf.clear()
def main(path):
    L = open(path).read().split("\n")
    out, buf, fence = [], [], False
    for l in L:
        if l.strip().startswith("```"):
            flush(buf, out); out.append(l); fence = not fence; continue
        if fence:
            out.append(l); continue
        if not l.strip() or l.startswith("#") or l.lstrip().startswith("|") or l.startswith(">") or re.match(r"^\s*(---+|===+)\s*$", l):
            flush(buf, out); out.append(l); continue
        if re.match(r"^\s*([*+-]|\d+\.)\s+", l):
            flush(buf, out); buf.append(l); continue
        # continuation line of a paragraph / list item
        if buf and re.match(r"^\s*([*+-]|\d+\.)\s+", buf[0]) and not l.startswith(" "):
            # an unindented line after a list item: new paragraph
            flush(buf, out)
        buf.append(l)
    flush(buf, out)
    open(path, "w").write("\n".join(out))
for p in sys.argv[1:]:
    main(p)
