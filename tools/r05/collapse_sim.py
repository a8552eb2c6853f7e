"""How many wide steps would a walk take with (a) the fixed two-level collapse of the binary tree (the quad records / own nodes of
today) and (b) a greedy surface-area collapse into 4-wide nodes (replace the internal child of largest area by its children until
four slots are full)?  CPU simulation on the binary SAH tree of the host builder (libythip's host side only: no GPU), exact float boxes,
children visited nearest-first, closest-hit rays approximated by the box test alone with a shrinking tmax = first leaf's entry
(leaf contents ignored: an upper-structure comparison, the same simplification on both sides)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "yocto-gl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ythip as yt, scenes as ysc
import parity as P

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_small"
nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 400
flat = {"cornell_small": lambda: P.scene_cornell_1m(n=60), "cornell1m": P.scene_cornell_1m, "hair": ysc.hair_scene, "plane": ysc.plane_scene}[name]()
bvh = yt.host_make_bvh(flat, True)
# the largest BLAS
sizes = np.diff(bvh.node_offset)
t = int(np.argmax(sizes[:-1]))
nodes = bvh.nodes[bvh.node_offset[t]:bvh.node_offset[t + 1]]
bmin, bmax = nodes["bbox_min"].astype(np.float64), nodes["bbox_max"].astype(np.float64)
internal, start = nodes["internal"].astype(bool), nodes["start"]
print(f"{name}: tree {t} with {len(nodes)} nodes ({internal.sum()} internal)")
area = lambda i: (lambda e: 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]))(bmax[i] - bmin[i])

def fixed(i):  # grandchildren (a child that is a leaf stays itself)
    out = []
    for c in (start[i], start[i] + 1):
        out += [start[c], start[c] + 1] if internal[c] else [c]
    return out

def greedy(i, width=4):
    slots = [start[i], start[i] + 1]
    while len(slots) < width:
        cand = [s for s in slots if internal[s]]
        if not cand: break
        s = max(cand, key=area)
        k = slots.index(s)
        slots[k:k + 1] = [start[s], start[s] + 1]
    return slots

def walk(children, o, d):
    inv = 1.0 / np.where(np.abs(d) < 1e-20, 1e-20, d)
    def box(i):
        a, b = (bmin[i] - o) * inv, (bmax[i] - o) * inv
        t0, t1 = np.minimum(a, b).max(), np.maximum(a, b).min()
        return max(t0, 1e-4), t1
    t0, t1 = box(0)
    if t0 > t1: return 0, 0
    stack, steps, leaves, tmax = [(t0, 0)], 0, 0, np.inf
    while stack:
        t0, i = stack.pop()
        if t0 > tmax: continue
        if not internal[i]:
            leaves += 1
            tmax = min(tmax, box(i)[1])  # (something in this leaf is hit at the latest when the ray leaves it: a crude closest-hit model)
            continue
        steps += 1
        hits = []
        for c in children(i):
            c0, c1 = box(c)
            if c0 <= c1 and c0 <= tmax: hits.append((c0, c))
        for h in sorted(hits, reverse=True): stack.append(h)
    return steps, leaves

rng = np.random.default_rng(7)
lo, hi = bmin[0], bmax[0]
slo, shi = flat.positions.min(0).astype(np.float64) - 0.25, flat.positions.max(0).astype(np.float64) + 0.25
res = {"fixed": [0, 0], "greedy": [0, 0]}
cache = {"fixed": {}, "greedy": {}}
def memo(kind, f):
    c = cache[kind]
    def g(i):
        if i not in c: c[i] = f(i)
        return c[i]
    return g
t0 = time.time()
for _ in range(nrays):
    o = slo + (shi - slo) * rng.random(3)                  # anywhere in the scene ...
    d = lo + (hi - lo) * rng.random(3) - o; d /= np.linalg.norm(d)   # ... towards a point of the shape's box
    for kind, f in (("fixed", fixed), ("greedy", greedy)):
        s, l = walk(memo(kind, f), o, d)
        res[kind][0] += s; res[kind][1] += l
for kind in res:
    used = cache[kind]
    fill = np.mean([len(v) for v in used.values()]) if used else 0
    print(f"  {kind:7s}: {res[kind][0] / nrays:7.2f} wide steps per ray, {res[kind][1] / nrays:6.2f} leaf visits, mean slots filled {fill:.2f} of 4")
print(f"  greedy / fixed steps = {res['greedy'][0] / max(res['fixed'][0], 1):.3f}   ({time.time() - t0:.1f} s)")
ids = np.flatnonzero(internal)[:20000]
diff = sum(1 for i in ids if sorted(fixed(i)) != sorted(greedy(i)))
lf = sum(1 for i in ids if len(fixed(i)) < 4); lg = sum(1 for i in ids if len(greedy(i)) < 4)
print(f"  of {len(ids)} internal nodes: greedy != fixed for {diff}; fewer than 4 slots: fixed {lf}, greedy {lg}")
