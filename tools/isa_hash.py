#!/usr/bin/env python3
"""Per-kernel hash of the ISA hipcc emits (labels and comments stripped): `isa_hash.py a.s [b.s]` prints the
hashes of a.s, or the kernels whose bodies differ between a.s and b.s.  Used to show that a refactor or a
diagnostic switch leaves the product kernels untouched."""
import hashlib, re, sys

def funcs(path):
    d, cur, buf = {}, None, []
    for l in open(path):
        if re.match(r'^\s*;|\s*\.file|\s*\.loc|\s*\.ident', l):
            continue
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur:
            if l.startswith('.Lfunc_end'):
                body = re.sub(r'(\.L[A-Za-z_]*|\bBB)\d+(_\d+)?', 'L', ''.join(buf))
                d[cur] = hashlib.md5(body.encode()).hexdigest()
                cur = None
            else:
                buf.append(l)
    return d

if __name__ == "__main__":
    a = funcs(sys.argv[1])
    if len(sys.argv) == 2:
        for k in sorted(a):
            print(a[k], k)
    else:
        b = funcs(sys.argv[2])
        diff = sorted(set(a) ^ set(b)) + [k for k in sorted(a) if k in b and a[k] != b[k]]
        print("%d kernels in %s, %d in %s, %d differ" % (len(a), sys.argv[1], len(b), sys.argv[2], len(diff)))
        for k in diff:
            print("  ", k)
        sys.exit(1 if diff else 0)
