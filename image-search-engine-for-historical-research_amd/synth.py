"""Counter-based synthetic descriptor generator (host side, numpy).

The reference ships no features, no ground truth and no weights, so every
test / bench input is synthetic.  Values are a pure function of (seed, row, col):

    ctr = row * D + col                       (uint64)
    h   = splitmix64(ctr + seed * 0xD1342543DE82EF95)
    v   = (sum of the four 16-bit fields of h  -  131070) * 2**-15      (float32, exact)

Only integer arithmetic and one exact power-of-two scaling are involved, so the numpy
generator here and the device generator in csrc/synth.hip (mi_synth_fill) produce
bit-identical float32 values, and any gallery shard can regenerate its own rows.
The sum of four uniforms is close to a normal with std 1.1547.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
SEED_MUL = 0xD1342543DE82EF95


def splitmix64(x):
    """SplitMix64 finaliser on a uint64 ndarray (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_rows(seed, row0, nrows, d, dtype=np.float32):
    """Rows [row0, row0+nrows) of the synthetic matrix with `d` columns -> [nrows, d]."""
    rows = np.arange(row0, row0 + nrows, dtype=np.uint64)[:, None]
    cols = np.arange(d, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ctr = rows * np.uint64(d) + cols
        off = np.uint64((int(seed) * SEED_MUL) & 0xFFFFFFFFFFFFFFFF)
        h = splitmix64(ctr + off)
    s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48)))
    v = (s.astype(np.int64) - 131070).astype(np.float32) * np.float32(2.0 ** -15)
    return v.astype(dtype, copy=False)


def planted_dataset(seed, n, d, nq, n_pos=(20, 60), sigmas=(0.35, 0.7, 1.1), dtype=np.float32):
    """Synthetic 'dataset' with planted neighbours and a revisited-style gnd.

    Returns (vecs[d, n], qvecs[d, nq], gnd) in the reference's column-per-image layout
    (src/networks/imageretrievalnet.py:370-379).  For query i a cluster of positives
    query + sigma * noise is written into disjoint gallery slots; the sigma band decides the
    easy / hard / junk label, giving a gnd list of dicts like the revisited pickles
    (src/datasets/testdataset.py:26-28).
    """
    g = synth_rows(seed, 0, n, d).astype(np.float64)
    q = synth_rows(seed + 1, 0, nq, d).astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    gnd = []
    rng_rows = 0
    slot = 0
    lo, hi = n_pos
    for i in range(nq):
        npos = lo + int(splitmix64(np.uint64(seed * 7919 + i)) % np.uint64(hi - lo + 1))
        labels = {"easy": [], "hard": [], "junk": []}
        for j in range(npos):
            if slot >= n:
                break
            band = j % 3
            noise = synth_rows(seed + 2, rng_rows, 1, d).astype(np.float64)[0]
            rng_rows += 1
            noise /= np.linalg.norm(noise)
            v = q[i] + sigmas[band] * noise
            g[slot] = v / np.linalg.norm(v)
            labels[("easy", "hard", "junk")[band]].append(slot)
            slot += 1
        gnd.append({k: np.array(v, dtype=np.int64) for k, v in labels.items()})
    return np.ascontiguousarray(g.T.astype(dtype)), np.ascontiguousarray(q.T.astype(dtype)), gnd
