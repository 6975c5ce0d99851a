"""CPU emulation of the speculative device build's index arithmetic (tree_build.hip kernels with
the guards), asserting that every read and write index stays in bounds also when the assumed
inside counts are wrong (duplicates / ties)."""
import numpy as np, sys
KD = 0xFFFFFFFF
def run(pts, capacity):
    n = len(pts)
    x = pts.copy(); row = np.arange(n); prov = np.zeros(n, np.uint64)
    level = [(0, n)] if n >= capacity else []
    remap = [0]
    mis = False
    depth = 0
    while level:
        S = len(level)
        segs = []; off = 0
        for (lo, hi) in level:
            assert hi - lo >= 2 or True
            segs.append((lo, hi, off)); off += hi - lo
        assert off <= n
        # keys
        seg = np.full(n, KD, np.uint64); d2 = np.zeros(n); keys = []
        for i in range(n):
            p = int(prov[i]); s = KD if p == KD else remap[p]
            assert p == KD or p < len(remap)
            seg[i] = s
            if s != KD:
                assert s < S
                lo, hi, _ = segs[s]; v = hi - 1; assert 0 <= v < n
                if i != v:
                    d2[i] = float(((x[v] - x[i]) ** 2).sum()); keys.append((s, d2[i]))
                else:
                    keys.append((s, np.inf))
            else:
                keys.append((S, 0.0))
        srt = sorted(keys)
        so = []
        for s, (lo, hi, off) in enumerate(segs):
            m = hi - lo - 1; idx = off + m // 2; assert 0 <= idx < n
            so.append(dict(r2=srt[idx][1], ni=0))
        flag = np.zeros(n, np.int64)
        for i in range(n):
            s = int(seg[i])
            if s != KD and i != segs[s][1] - 1:
                flag[i] = 1 if d2[i] < so[s]['r2'] else 0
        scan = np.concatenate([[0], np.cumsum(flag)[:-1]])
        for s, (lo, hi, off) in enumerate(segs):
            so[s]['ni'] = int(scan[hi - 1] - scan[lo])
        ox = x.copy(); oprov = np.full(n, KD, np.uint64); best = [0] * (2 * S)
        for i in range(n):
            s = int(seg[i]); pos = i; pr = KD
            if s != KD:
                lo, hi, _ = segs[s]
                if i != hi - 1 and lo <= i < hi:
                    ni = so[s]['ni']; rank_in = int(scan[i] - scan[lo]); assert rank_in >= 0
                    inn = flag[i] != 0
                    local = rank_in if inn else (i - lo - rank_in)
                    assert local >= 0
                    pos = lo + local if inn else lo + ni + local
                    child = 2 * s + (0 if inn else 1); pr = child
                    key = (d2[i], 0xFFFFFFFF - local)
                    if best[child] == 0 or key > best[child]: best[child] = key
            if pos >= n:
                pos = i; pr = KD
            assert 0 <= pos < n, (pos, n)
            ox[pos] = x[i]; oprov[pos] = pr
        for c in range(2 * S):
            lo0, hi0, _ = segs[c >> 1]; ni = so[c >> 1]['ni']
            lo = lo0 + ni if c & 1 else lo0; hi = hi0 - 1 if c & 1 else lo0 + ni
            if hi <= lo: continue
            k = 0xFFFFFFFF - (best[c][1] if best[c] != 0 else 0)
            if k >= hi - lo: continue
            a = lo + k; b = hi - 1; assert 0 <= a < n and 0 <= b < n
        x = ox; prov = oprov
        nxt = []; remap = [KD] * (2 * S)
        for s, (lo, hi, off) in enumerate(segs):
            m = hi - lo - 1; ni_pred = m // 2
            if so[s]['ni'] != ni_pred: mis = True
            ni = ni_pred
            if ni == 0 or m - ni == 0: return 'degenerate-plan', mis
            for side, (clo, chi) in enumerate(((lo, lo + ni), (lo + ni, hi - 1))):
                if chi - clo >= capacity:
                    remap[2 * s + side] = len(nxt); nxt.append((clo, chi))
        for (lo, hi) in nxt:
            if hi - lo < 2: return 'tiny', mis
        level = nxt; depth += 1
    return depth, mis
rng = np.random.default_rng(0)
for trial in range(30):
    n = int(rng.integers(10, 400)); cap = int(rng.choice([3, 4, 8, 16]))
    kind = trial % 3
    if kind == 0: pts = rng.normal(size=(n, 3))
    elif kind == 1: pts = np.repeat(rng.normal(size=(max(1, n // 5), 3)), 5, axis=0)[:n]   # duplicates
    else: pts = np.zeros((n, 3))                                                           # all equal
    print(trial, n, cap, kind, run(pts.astype(np.float32).astype(np.float64), cap))
