#!/usr/bin/env python3
"""Builds, verifies and emits the min / max network of the 5 x 5 median kernel (image-processing_amd/csrc/ipx_median5_net.inc).

The median of a 5 x 5 window (cv2.medianBlur(gray, 5), server.py:436 -- denoise_manual's default) as three pieces that
neighbouring windows share:
  A) med5_sort5:  sort the 5 vertical values of a column (9 comparators; a column belongs to 5 windows),
  B) med5_merge:  Batcher odd-even merge of two adjacent sorted columns into a sorted 10-list (13 comparators; the kernel
                  only merges the pairs that start at an even column, each of which serves 4 windows),
  C) med5_middle: ranks 7..12 of the union of two such 10-lists (the odd-even merge restricted to what they need; 7 of
                  a window's 25 values are then known to lie below them and 7 above).  Four adjacent columns belong to
                  two windows, the one that adds the column to their right and the one that adds the column to their left,
  D) med5_final:  per window, rank 5 of those six and the window's fifth sorted column.
The whole network is checked here on all 2^25 0-1 inputs (bit-parallel on Python integers; by the 0-1 principle a
min / max network that selects the median of every 0-1 input selects it for every input), the unused halves of the
comparators of C and D are dropped, and the four pieces are written as C++ over the macros M5_T / M5_MIN / M5_MAX, so that
tests/test_median5_net.py can run the EMITTED text through the same exhaustive check on the CPU (M5_T = uint64_t,
and / or) before the kernel uses it on packed 16-bit pairs.

    python scripts/gen_median5_net.py            # verify and rewrite the .inc
    python scripts/gen_median5_net.py --check    # verify, and fail if the committed .inc differs
"""
import os
import sys

SORT5 = [(0, 1), (3, 4), (2, 4), (2, 3), (1, 4), (0, 3), (0, 2), (1, 3), (1, 2)]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-processing_amd", "csrc", "ipx_median5_net.inc")


def oddeven_merge(a, b):
    """Batcher's odd-even merge of two sorted wire lists of any lengths -> (comparators, wires in merged order).
    A comparator (i, j) leaves the minimum on wire i and the maximum on wire j."""
    if not a:
        return [], list(b)
    if not b:
        return [], list(a)
    if len(a) == 1 and len(b) == 1:
        return [(a[0], b[0])], [a[0], b[0]]
    ce1, ev = oddeven_merge(a[0::2], b[0::2])
    ce2, od = oddeven_merge(a[1::2], b[1::2])
    ces = ce1 + ce2
    out = [ev[0]]
    k = 0
    while k < len(od) and k + 1 < len(ev):                     # Knuth 5.3.4: ev0, then (od_k, ev_k+1) pairs, leftovers in order
        ces.append((od[k], ev[k + 1]))
        out += [od[k], ev[k + 1]]
        k += 1
    out += od[k:] + ev[k + 1:]
    return ces, out


def inputs01(n):
    """Wire i as the integer whose bit p is bit i of p, over all 2^n patterns p."""
    N = 1 << n
    w = []
    for i in range(n):
        x = ((1 << (1 << i)) - 1) << (1 << i)                  # 2^i zeros then 2^i ones
        width = 1 << (i + 1)
        while width < N:
            x |= x << width
            width <<= 1
        w.append(x)
    return w


def run(ces, w):
    for (i, j) in ces:
        a, b = w[i], w[j]
        w[i], w[j] = a & b, a | b
    return w


def at_least(n, k):
    """Integer whose bit p is 1 iff popcount(p) >= k, p < 2^n."""
    N = 1 << n
    nb = 6
    cnt = [0] * nb
    for x in inputs01(n):                                      # bit-sliced ripple counter
        c = x
        for b in range(nb):
            cnt[b], c = cnt[b] ^ c, cnt[b] & c
    full = (1 << N) - 1
    ge, eq = 0, full
    for b in reversed(range(nb)):
        if (k >> b) & 1:
            eq &= cnt[b]
        else:
            ge |= eq & cnt[b]
            eq &= ~cnt[b] & full
    return ge | eq


def prune(ces, outs):
    """Drop what the wires in `outs` do not depend on -> [(i, j, min needed, max needed)]."""
    need = set(outs)
    kept = []
    for (i, j) in reversed(ces):
        ni, nj = i in need, j in need
        if ni or nj:
            kept.append((i, j, ni, nj))
            need.add(i)
            need.add(j)
    kept.reverse()
    return kept


def pieces():
    """The three pieces over their own wire numbering."""
    a, b = list(range(5)), list(range(5, 10))
    merge_ces, merge_out = oddeven_merge(a, b)
    p, q, c = list(range(10)), list(range(10, 20)), list(range(20, 25))
    m20, o20 = oddeven_merge(p, q)
    mid = o20[7:13]
    m11, o11 = oddeven_merge(mid, c)
    fin = prune(m11, [o11[5]])
    need = {k for i, j, _, _ in fin for k in (i, j)} & set(mid)           # what D reads of the six
    return (merge_ces, merge_out), (prune(m20, sorted(need)), mid), (fin, o11[5])


def verify(merge, middle, final):
    """The composed network (5 column sorts, 2 pair merges, the middle six, the final selection) on all 2^25 0-1 inputs."""
    (merge_ces, merge_out), (mid_ces, mid), (fin, out) = merge, middle, final
    sel = mid_ces + fin
    w = inputs01(25)                                           # wire 5 c + r = column c, row r
    for c in range(5):
        run([(5 * c + i, 5 * c + j) for i, j in SORT5], w)
    lists = []
    for c in (0, 2):
        m = {k: 5 * c + k for k in range(10)}
        run([(m[i], m[j]) for i, j in merge_ces], w)
        lists.append([m[k] for k in merge_out])
    v = [w[k] for k in lists[0]] + [w[k] for k in lists[1]] + w[20:25]
    for (i, j, ni, nj) in sel:
        a, b = v[i], v[j]
        v[i] = a & b if ni else None
        v[j] = a | b if nj else None
    return v[out] == at_least(25, 13)


def emit_ce(i, j, ni, nj, name):
    a, b = name(i), name(j)
    if ni and nj:
        return f"    {{ const M5_T lo = M5_MIN({a}, {b}); {b} = M5_MAX({a}, {b}); {a} = lo; }}"
    if ni:
        return f"    {a} = M5_MIN({a}, {b});"
    return f"    {b} = M5_MAX({a}, {b});"


def emit(merge, middle, final):
    (merge_ces, merge_out), (mid_ces, mid), (fin, out) = merge, middle, final
    nops = lambda ces: sum(int(a) + int(b) for _, _, a, b in ces)
    L = []
    L.append("// ipx_median5_net.inc -- GENERATED by scripts/gen_median5_net.py (which verifies it on all 2^25 0-1 inputs); do not edit.")
    L.append("// Needs M5_T (value type), M5_MIN(a, b), M5_MAX(a, b) and M5_FN (function qualifiers).")
    L.append(f"// Operations: sort5 {2 * len(SORT5)}, merge {2 * len(merge_ces)}, middle {nops(mid_ces)}, final {nops(fin)} (min or max each).")
    L.append("")
    L.append("// v[0] <= v[1] <= ... <= v[4] afterwards")
    L.append("M5_FN void med5_sort5(M5_T (&v)[5])")
    L.append("{")
    for i, j in SORT5:
        L.append(emit_ce(i, j, True, True, lambda k: f"v[{k}]"))
    L.append("}")
    L.append("")
    L.append("// a and b sorted -> p = all ten, sorted")
    L.append("M5_FN void med5_merge(const M5_T (&a)[5], const M5_T (&b)[5], M5_T (&p)[10])")
    L.append("{")
    L.append("    M5_T " + ", ".join(f"w{k} = {'a' if k < 5 else 'b'}[{k % 5}]" for k in range(10)) + ";")
    for i, j in merge_ces:
        L.append(emit_ce(i, j, True, True, lambda k: f"w{k}"))
    for r, k in enumerate(merge_out):
        L.append(f"    p[{r}] = w{k};")
    L.append("}")
    L.append("")
    L.append("// p, q sorted 10-lists -> m = ranks 7 .. 12 of the twenty, sorted")
    L.append("M5_FN void med5_middle(const M5_T (&p)[10], const M5_T (&q)[10], M5_T (&m)[6])")
    L.append("{")
    used = sorted({k for i, j, _, _ in mid_ces for k in (i, j)})
    L.append("    M5_T " + ", ".join(f"w{k} = {'p' if k < 10 else 'q'}[{k % 10}]" for k in used) + ";")
    for i, j, ni, nj in mid_ces:
        L.append(emit_ce(i, j, ni, nj, lambda k: f"w{k}"))
    for r, k in enumerate(mid):
        L.append(f"    m[{r}] = w{k};")
    L.append("}")
    L.append("")
    L.append("// m from med5_middle and c the window's fifth sorted column -> the median of the 25 (rank 12 from 0)")
    L.append("M5_FN M5_T med5_final(const M5_T (&m)[6], const M5_T (&c)[5])")
    L.append("{")
    used = sorted({k for i, j, _, _ in fin for k in (i, j)})
    src = lambda k: f"m[{mid.index(k)}]" if k in mid else f"c[{k - 20}]"
    L.append("    M5_T " + ", ".join(f"w{k} = {src(k)}" for k in used) + ";")
    for i, j, ni, nj in fin:
        L.append(emit_ce(i, j, ni, nj, lambda k: f"w{k}"))
    L.append(f"    return w{out};")
    L.append("}")
    return "\n".join(L) + "\n"


if __name__ == "__main__":
    merge, middle, final = pieces()
    ok = verify(merge, middle, final)
    print("median of 25 on all 2^25 0-1 inputs:", "correct" if ok else "WRONG")
    if not ok:
        sys.exit(1)
    text = emit(merge, middle, final)
    print(text.split("\n")[2][3:])
    if "--check" in sys.argv:
        same = os.path.exists(OUT) and open(OUT).read() == text
        print("committed file is", "up to date" if same else "DIFFERENT")
        sys.exit(0 if same else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", os.path.relpath(OUT))
