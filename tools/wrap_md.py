"""Re-flow the prose of a markdown file to a line width (tables, headings, fenced code and paragraphs that already fit are left alone):
python3 tools/wrap_md.py FILE [width].  A paragraph (or one bullet's text) with an over-long line is joined and cut again at spaces; continuation
lines keep the indentation the paragraph had (two more than the marker under a bullet), which markdown reads as the same paragraph."""
import re
import sys

BAD_START = re.compile(r"^([-+*>#|]|\d+[.)])$")
BULLET = re.compile(r"^(\s*)([-*+]|\d+[.)])\s+")


def flow(group, width):
    first = group[0]
    lead = re.match(r"^\s*", first).group(0)
    b = BULLET.match(first)
    if b:
        indent = b.group(1) + "  "
    elif len(group) > 1:
        indent = re.match(r"^\s*", group[1]).group(0)
    else:
        indent = lead
    words = " ".join(l.strip() for l in group).split(" ")
    out, cur = [], lead + words[0]
    for w in words[1:]:
        if len(cur) + 1 + len(w) > width and not BAD_START.match(w):
            out.append(cur)
            cur = indent + w
        else:
            cur += " " + w
    out.append(cur)
    return out


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    res, group, fenced = [], [], False

    def close():
        if group:
            res.extend(flow(group, width) if any(len(l) > width for l in group) else group)
            group.clear()

    for line in open(path).read().split("\n"):
        s = line.lstrip()
        if s.startswith("```"):
            close()
            fenced = not fenced
            res.append(line)
        elif fenced or s == "" or s.startswith(("|", "#")):
            close()
            res.append(line)
        else:
            if BULLET.match(line):
                close()
            group.append(line)
    close()
    open(path, "w").write("\n".join(res))


if __name__ == "__main__":
    main()
