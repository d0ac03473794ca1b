"""Crude structure check of the Julia shim (no julia in the image): strings / comments stripped, block openers against `end`, ()/[] depth.
usage: python scripts/jl_balance.py finitediff.jl_amd/julia/FiniteDiffMI355X.jl"""
import re, sys
src = open(sys.argv[1]).read()
# strip block comments, strings, line comments, chars
out = []
i = 0
n = len(src)
while i < n:
    c = src[i]
    if src.startswith('#=', i):
        j = src.find('=#', i + 2); i = n if j < 0 else j + 2; continue
    if c == '#':
        j = src.find('\n', i); i = n if j < 0 else j; continue
    if src.startswith('"""', i):
        j = src.find('"""', i + 3); out.append('""'); i = n if j < 0 else j + 3; continue
    if c == '"':
        j = i + 1
        while j < n and src[j] != '"':
            if src[j] == '\\': j += 1
            j += 1
        out.append('""'); i = j + 1; continue
    if c == "'" and i + 2 < n and (src[i+2] == "'" or (src[i+1] == '\\' and i + 3 < n and src[i+3] == "'")):
        j = src.find("'", i + 1 if src[i+1] != '\\' else i + 3); out.append("' '"); i = j + 1; continue
    out.append(c); i += 1
txt = ''.join(out)
openers = {'function', 'if', 'for', 'while', 'struct', 'module', 'baremodule', 'let', 'do', 'try', 'begin', 'quote', 'macro'}
depth_b = 0  # [] depth
depth_p = 0  # () depth
stack = []
line = 1
for m in re.finditer(r'[A-Za-z_!][A-Za-z_0-9!]*|\n|[\[\]\(\)\{\}]|:[A-Za-z_]+', txt):
    t = m.group(0)
    if t == '\n': line += 1; continue
    if t == '[' or t == '{': depth_b += 1; continue
    if t == ']' or t == '}': depth_b -= 1; continue
    if t == '(': depth_p += 1; continue
    if t == ')': depth_p -= 1; continue
    if t.startswith(':'): continue   # symbols like :end, :if
    if depth_b > 0: continue          # indexing / comprehensions
    if t in openers:
        if t in ('for', 'if') and depth_p > 0: continue     # generators inside parentheses
        # `mutable struct`, `abstract type ... end`, `primitive type` -- type handled below
        stack.append((t, line)); continue
    if t == 'type' :
        # abstract type / primitive type open a block
        prev = txt[max(0, m.start() - 12):m.start()]
        if re.search(r'(abstract|primitive)\s+$', prev): stack.append((t, line))
        continue
    if t == 'end':
        if not stack: print("unmatched end at line", line); sys.exit(1)
        stack.pop()
print("depth () %d [] %d, unclosed blocks: %s" % (depth_p, depth_b, stack[-5:]))
