import re, sys
big = []
for ln in open(sys.argv[1], errors="replace"):
    for m in re.finditer(r"(\d+\.\d+) ms", ln):
        if float(m.group(1)) >= float(sys.argv[2]):
            big.append(ln.rstrip()[:260]); break
print(len(big), "lines with a figure >= %s ms" % sys.argv[2])
for b in big[:60]: print(b)
