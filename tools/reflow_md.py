"""Re-wrap the prose of a Markdown file at a diffable width (tables, code fences and headings are left alone):
    python tools/reflow_md.py FILE [width=124]"""
import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)([-*]|\d+\.)\s+")


def flush(buf, out, width):
    if not buf:
        return
    first = buf[0]
    m = ITEM.match(first)
    if m:
        indent = m.group(1)
        marker = first[len(indent):m.end()]
        body = " ".join([first[m.end():].strip()] + [b.strip() for b in buf[1:]])
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent + marker, subsequent_indent=indent + " " * len(marker), break_long_words=False, break_on_hyphens=False))
    else:
        indent = re.match(r"^\s*", first).group(0)
        body = " ".join(b.strip() for b in buf)
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    buf.clear()


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 124
    out, buf, fence = [], [], False
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            flush(buf, out, width)
            fence = not fence
            out.append(line)
        elif fence or line.startswith("|") or line.startswith("#") or not line.strip():
            flush(buf, out, width)
            out.append(line)
        elif ITEM.match(line):
            flush(buf, out, width)
            buf.append(line)
        else:
            buf.append(line)
    flush(buf, out, width)
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
