#!/usr/bin/env python3
"""Re-wraps the prose paragraphs of a Markdown file to a line width (default 118) without touching tables, code fences,
headings or blank lines; list items keep their hanging indent.  (DESIGN.md had 1 000-character lines.)

    python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    lines = open(path).read().split("\n")
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([-*]|\d+\.)\s+", first)
        if m:
            indent = m.group(0)
            body = first[len(indent):] + " " + " ".join(x.strip() for x in para[1:])
            out.extend(textwrap.wrap(body.strip(), width=width, initial_indent=indent, subsequent_indent=" " * len(indent),
                                     break_long_words=False, break_on_hyphens=False))
        else:
            lead = re.match(r"^\s*", first).group(0)
            body = " ".join(x.strip() for x in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=lead, break_long_words=False,
                                     break_on_hyphens=False))
        para.clear()

    for ln in lines:
        if ln.strip().startswith("```"):
            flush()
            fence = not fence
            out.append(ln)
            continue
        if fence or not ln.strip() or ln.lstrip().startswith(("|", "#", ">")) or re.match(r"^\s{4,}\S", ln) and not para:
            flush()
            out.append(ln)
            continue
        if re.match(r"^\s*([-*]|\d+\.)\s+", ln) and para:      # a new list item ends the previous paragraph
            flush()
        para.append(ln)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
