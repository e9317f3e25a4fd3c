"""numpy restatement of the counter-based PLAN engine (``MSIM_RNG_FAST`` / ``--rng fast``), written from the description
in ``csrc/fast_math.h`` -- Philox4x32-10 counters, the splitting tree of hypergeometric draws, the leaves' rejection sampling,
the per-candidate draws -- with the boundary pass and the visit filter the way the REFERENCE writes them (mutator.py:184-213,
318-421: sequential loops).  Test infrastructure: the device (``tests/test_gpu_fast_rng.py``) and the library's own sequential
restatement (``msim_dbg_fast_plan``, ``tests/test_fast_host.py``) must reproduce it bit for bit.

Everything floating point is IEEE double +, -, *, / in a fixed order (numpy never contracts), so the results are exact
twins of the C++ / HIP code built with contraction off."""
from __future__ import annotations

import numpy as np

U64 = np.uint64
M32 = U64(0xFFFFFFFF)
TAG_SPLIT, TAG_POS, TAG_CAND, TAG_INS = 16, 17, 18, 19
LG_MIN, LG_MAX, LEAF_TARGET = 10, 16, 224
SN, IN, DE, DU, IV = 1, 2, 3, 4, 5


def _u64(x):
    return np.asarray(x, dtype=np.uint64)


def philox(c0, c1, c2, c3, key):
    """Philox4x32-10 (Salmon et al. 2011) over uint64 arrays holding 32-bit values; key = 64-bit int."""
    c = [_u64(x) for x in np.broadcast_arrays(_u64(c0), _u64(c1), _u64(c2), _u64(c3))]
    k0, k1 = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = U64(0xD2511F53) * c[0], U64(0xCD9E8D57) * c[2]
        c = [(p1 >> U64(32)) ^ c[1] ^ U64(k0), p1 & M32, (p0 >> U64(32)) ^ c[3] ^ U64(k1), p0 & M32]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def draw4(key, seq, x, y, tag, hi=0):
    w = _u64(tag) | (_u64(hi) << U64(8))
    return philox(x, y, seq, w, key)


def lo64(v):
    return (v[1] << U64(32)) | v[0]


def hi64(v):
    return (v[3] << U64(32)) | v[2]


def mulhi64(a, b):
    a, b = np.broadcast_arrays(_u64(a), _u64(b))
    al, ah, bl, bh = a & M32, a >> U64(32), b & M32, b >> U64(32)
    ll, lh, hl, hh = al * bl, al * bh, ah * bl, ah * bh
    mid = (ll >> U64(32)) + (lh & M32) + (hl & M32)
    return hh + (lh >> U64(32)) + (hl >> U64(32)) + (mid >> U64(32))


# ---------------------------------------------------------------------------------------------- IEEE-only helpers
def uni52(r64):
    return ((_u64(r64) >> U64(12)).astype(np.float64) + 0.5) * (1.0 / 4503599627370496.0)


def d_log(x):
    x = np.asarray(x, dtype=np.float64)
    b = x.view(np.uint64) if x.ndim else np.array([x]).view(np.uint64)
    b = b.reshape(x.shape)
    e = ((b >> U64(52)) & U64(0x7FF)).astype(np.int64) - 1023
    f = ((b & U64(0x000FFFFFFFFFFFFF)) | U64(0x3FF0000000000000)).view(np.float64)
    big = f > 1.4142135623730951
    f = np.where(big, f * 0.5, f)
    e = e + big
    s = (f - 1.0) / (f + 1.0)
    z = s * s
    p = np.full(x.shape, 1.0 / 23.0)
    for q in (21.0, 19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * z + 1.0 / q
    p = p * z + 1.0
    t = (2.0 * s) * p
    return e.astype(np.float64) * 0.6931471805599453 + t


def _atanh_series(s):
    z = s * s
    p = np.full(np.shape(s), 1.0 / 23.0)
    for q in (21.0, 19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * z + 1.0 / q
    p = p * z + 1.0
    return (2.0 * s) * p


def d_log1p(t):
    t = np.asarray(t, dtype=np.float64)
    near = (t > -0.25) & (t < 0.4)
    tn = np.where(near, t, 0.0)
    series = _atanh_series(tn / (2.0 + tn))
    far = d_log(np.where(near, 1.0, 1.0 + t))
    return np.where(near, series, far)


def d_sqrt_up(x):
    x = np.asarray(x, dtype=np.float64)
    b = x.view(np.uint64).reshape(x.shape)
    e = ((b >> U64(52)) & U64(0x7FF)).astype(np.int64) - 1023
    y = ((1023 + (e >> 1)).astype(np.uint64) << U64(52)).view(np.float64)
    for _ in range(4):
        y = 0.5 * (y + x / y)
    return y


def stirling_corr_inv(iy):
    iy2 = iy * iy
    return iy * (1.0 / 12.0 - iy2 * (1.0 / 360.0 - iy2 * (1.0 / 1260.0)))


def log_factorial(x):
    x = np.asarray(x, dtype=np.int64)
    small = x < 32
    p = np.ones(x.shape)
    for i in range(2, 32):
        p = np.where(small & (x >= i), p * float(i), p)
    y = np.where(small, 33.0, (x + 1).astype(np.float64))
    big = (y - 0.5) * d_log(y) - y + 0.9189385332046727 + stirling_corr_inv(1.0 / y)
    return np.where(small, d_log(p), big)


def log_factorial_diff(a, d):
    a = np.asarray(a, dtype=np.int64)
    d = np.asarray(d, dtype=np.int64)
    a, d = np.broadcast_arrays(a, d)
    a1 = a + d
    stable = (a + 1 >= 32) & (a1 + 1 >= 32)
    y0 = np.where(stable, a + 1, 40).astype(np.float64)
    y1 = np.where(stable, a1 + 1, 40).astype(np.float64)
    dd = np.where(stable, d, 0).astype(np.float64)
    iy0 = 1.0 / y0
    st = (y1 - 0.5) * d_log1p(dd * iy0) + dd * (d_log(y0) - 1.0) + (stirling_corr_inv(1.0 / y1) - stirling_corr_inv(iy0))
    plain = log_factorial(np.where(stable, 0, a1)) - log_factorial(np.where(stable, 0, a))
    return np.where(d == 0, 0.0, np.where(stable, st, plain))


# ---------------------------------------------------------------------------------------------- hypergeometric
def hypergeometric(good, bad, sample, key, seq, node, rng):
    """Vectorised twin of fast_math.h: hypergeometric().  All arguments broadcastable integer arrays."""
    good, bad, sample, node = [np.asarray(v, dtype=np.int64) for v in np.broadcast_arrays(good, bad, sample, node)]
    out = np.zeros(good.shape, dtype=np.int64)
    N = good + bad
    triv0 = (sample == 0) | (good == 0)
    triv1 = ~triv0 & (bad == 0)
    triv2 = ~triv0 & ~triv1 & (sample >= N)
    out[triv1] = sample[triv1]
    out[triv2] = good[triv2]
    live = ~(triv0 | triv1 | triv2)
    m = np.minimum(sample, N - sample)
    z = np.zeros(good.shape, dtype=np.int64)
    urn = live & (m <= 10)
    if urn.any():
        ix = np.flatnonzero(urn)
        g, n, cnt, mm = good.flat[ix].copy(), N.flat[ix].copy(), np.zeros(len(ix), np.int64), m.flat[ix]
        nd = node.flat[ix]
        v = None
        for i in range(10):
            if not (i & 1):
                v = draw4(key, seq, nd, i >> 1, TAG_SPLIT, rng)
            r = hi64(v) if (i & 1) else lo64(v)
            act = i < mm
            hit = act & (mulhi64(r, np.where(act, n, 1)).astype(np.int64) < g)
            g = g - hit
            cnt = cnt + hit
            n = n - act
        z.flat[ix] = cnt
    hr = live & (m > 10)
    if hr.any():
        ix = np.flatnonzero(hr)
        g_, b_, m_, N_, nd = good.flat[ix], bad.flat[ix], m.flat[ix], N.flat[ix], node.flat[ix]
        mingb, maxgb = np.minimum(g_, b_), np.maximum(g_, b_)
        p = mingb.astype(np.float64) / N_.astype(np.float64)
        q = 1.0 - p
        a = m_.astype(np.float64) * p + 0.5
        var = (N_ - m_).astype(np.float64) * m_.astype(np.float64) * p * q / (N_ - 1).astype(np.float64)
        c = d_sqrt_up(var + 0.5)
        h = 1.7155277699214135 * c + 0.8989161620588988
        mode = ((m_ + 1).astype(object) * (mingb + 1).astype(object) // (N_ + 2).astype(object)).astype(np.int64)
        lim_a = (np.minimum(m_, mingb) + 1).astype(np.float64)
        lim_b = np.floor(a + 16.0 * c)
        bnd = np.where(lim_a < lim_b, lim_a, lim_b)
        Z = mode.copy()
        todo = np.ones(len(ix), dtype=bool)
        att = 0
        while todo.any():
            assert att < 4096
            t = np.flatnonzero(todo)
            v = draw4(key, seq, nd[t], att, TAG_SPLIT, rng)
            U, V = uni52(lo64(v)), uni52(hi64(v))
            X = a[t] + h[t] * (V - 0.5) / U
            ok = ~((X < 0.0) | (X >= bnd[t]))
            Zc = np.where(ok, X, 0.0).astype(np.int64)
            dz = Zc - mode[t]
            T = -(log_factorial_diff(mode[t], np.where(ok, dz, 0)) + log_factorial_diff(mingb[t] - mode[t], np.where(ok, -dz, 0)) +
                  log_factorial_diff(m_[t] - mode[t], np.where(ok, -dz, 0)) + log_factorial_diff(maxgb[t] - m_[t] + mode[t], np.where(ok, dz, 0)))
            acc1 = U * (4.0 - U) - 3.0 <= T
            rej2 = U * (U - T) >= 1.0
            acc3 = 2.0 * d_log(U) <= T
            acc = ok & (acc1 | (~rej2 & acc3))
            Z[t[acc]] = Zc[acc]
            todo[t[acc]] = False
            att += 1
        z.flat[ix] = np.where(g_ > b_, m_ - Z, Z)
    res = np.where(m < sample, good - z, z)
    out[live] = res[live]
    return out


# ---------------------------------------------------------------------------------------------- positions
def leaf_lg(n, k):
    e = LG_MIN
    while e < LG_MAX and (k << e) < LEAF_TARGET * n:
        e += 1
    return e


def tree_counts(n, k, lgB, key, seq, r):
    """points per leaf of drawing range r (n values, k points, leaves of 2^lgB values)."""
    B = 1 << lgB
    T = (n + B - 1) >> lgB
    lgP = 0
    while (1 << lgP) < T:
        lgP += 1
    m = np.zeros(1 << lgP, dtype=np.int64)
    m[0] = k
    for lev in range(lgP):
        S = 1 << (lgP - lev)
        half = S >> 1
        i = np.arange(1 << lev, dtype=np.int64)
        a = i * S
        mid = a + half
        act = mid < T
        a, mid, i = a[act], mid[act], i[act]
        if not len(a):
            continue
        Kn = m[a]
        va, vm = a << lgB, mid << lgB
        vb = np.minimum((a + S) << lgB, n)
        kl = hypergeometric(vm - va, vb - vm, Kn, key, seq, (1 << lev) + i, r)
        m[a] = kl
        m[mid] = Kn - kl
    return m[:T]


def leaf_values(key, seq, leaf, length, m):
    """the m values of one leaf, ascending."""
    if m == 0:
        return np.zeros(0, dtype=np.int64)
    inv = 2 * m > length
    need = length - m if inv else m
    have = np.zeros(0, dtype=np.int64)
    J = 0
    firsts = {}
    while True:
        J2 = J + max(64, 2 * (need - len(firsts)) + 64)
        j = np.arange(J, J2, dtype=np.uint64)
        v4 = draw4(key, seq, j >> U64(1), leaf, TAG_POS)
        r = np.where((j & U64(1)) == 1, hi64(v4), lo64(v4))
        vals = mulhi64(r, length).astype(np.int64)
        for val in vals.tolist():                      # the first `need` distinct values of the sequence
            if val not in firsts:
                firsts[val] = True
                if len(firsts) == need:
                    break
        if len(firsts) == need:
            break
        J = J2
    have = np.array(sorted(firsts), dtype=np.int64)
    if inv:
        mask = np.ones(length, dtype=bool)
        mask[have] = False
        have = np.flatnonzero(mask).astype(np.int64)
    return have


def range_positions(start, stop, k, d, key, seq, r, leaf_base):
    n = (stop - (k - 1) * d) - start
    lgB = leaf_lg(n, k)
    m = tree_counts(n, k, lgB, key, seq, r)
    B = 1 << lgB
    vals = []
    for t, mt in enumerate(m.tolist()):
        if mt:
            vals.append((t << lgB) + leaf_values(key, seq, leaf_base + t, min(B, n - (t << lgB)), mt))
    v = np.concatenate(vals) if vals else np.zeros(0, dtype=np.int64)
    assert len(v) == k
    return start + v + d * np.arange(k, dtype=np.int64), len(m)


# ---------------------------------------------------------------------------------------------- the whole plan
def plan(L, ranges, block, ti_lim, key, seq):
    """ranges: dicts {start, stop, k, types[], thr[], min_len{t}, max_len{t}}; block: {t: block}.  Returns (records, pool)
    with records as a list of (pos, stop, extra, type, aux) in position order -- what msim_plan_contig leaves in a fast context."""
    d = min(block[t] for t in range(1, 8))
    b1 = {t: min(block[t] + 1, 0xFFFFFFFF) for t in range(1, 8)}
    kept = []                                           # (pos, stop, ord, type)
    ord0 = 0
    leaf_base = 0
    r_i = 0
    for rg in ranges:
        k = rg["k"]
        if k == 0:
            continue
        pos, n_leaves = range_positions(rg["start"], rg["stop"], k, d, key, seq, r_i, leaf_base)
        ords = ord0 + np.arange(k, dtype=np.int64)
        v = draw4(key, seq, ords, 0, TAG_CAND)
        u53 = lo64(v) >> U64(11)
        thr = np.array(rg["thr"], dtype=np.uint64)
        idx = (thr[None, :] <= u53[:, None]).sum(axis=1)
        idx = np.minimum(idx, len(thr) - 1)
        types = np.array(rg["types"], dtype=np.int64)[idx]
        r64 = hi64(v)
        stop = pos.copy()
        dropped = np.zeros(k, dtype=bool)
        for t in (IN, DE, DU, IV):
            sel = types == t
            if not sel.any():
                continue
            lo, hi = rg["min_len"][t], rg["max_len"][t]
            st = pos[sel] + lo - 1 + mulhi64(r64[sel], hi - lo + 1).astype(np.int64)
            if t == IV:
                drop = pos[sel] + hi >= L - 1
                st = np.where(drop, pos[sel], st)
                dropped[np.flatnonzero(sel)[drop]] = True
            elif t != IN:
                st = np.minimum(st, L - 1)
            stop[sel] = st
        blocked_end = 0                                 # last_mut_range = range(0)          mutator.py:184
        for p_, s_, t_, o_, dr in zip(pos.tolist(), stop.tolist(), types.tolist(), ords.tolist(), dropped.tolist()):
            if p_ < blocked_end:                        # mutator.py:190-192
                continue
            if dr:                                      # mutator.py:199-201
                continue
            blocked_end = (p_ if t_ in (SN, IN) else s_) + b1[t_]
            kept.append((p_, s_, o_, t_))
        ord0 += k
        leaf_base += n_leaves
        r_i += 1
    vis = []
    cover = -1
    for p_, s_, o_, t_ in kept:
        if p_ <= cover:                                 # inside an earlier DE / DU / IV span: never visited   mutator.py:376,386,398
            continue
        if t_ in (DE, DU, IV):
            cover = s_
        vis.append((p_, s_, o_, t_))
    if not vis:
        return [], b"", len(kept) == 0
    P, S, O, T = (np.array(c, dtype=np.int64) for c in zip(*vis))
    aux = np.zeros(len(P), dtype=np.int64)
    sn = T == SN
    if sn.any():                                        # mutator.py:428-455: the high 64 bits of the candidate's draw
        r = hi64(draw4(key, seq, O[sn], 0, TAG_CAND))
        aux[sn] = np.where((r >> U64(11)) < U64(ti_lim), 0, 1 + (r & U64(1)).astype(np.int64))
    extra = np.zeros(len(P), dtype=np.int64)
    ins = T == IN
    pool = np.zeros(0, dtype=np.uint8)
    if ins.any():                                       # mutator.py:465-471: 64 bases per counter, 2 bits each
        ln = (S - P + 1)[ins]
        off = np.concatenate(([0], np.cumsum(ln)))
        extra[ins] = off[:-1]
        rec_of = np.repeat(np.arange(len(ln)), ln)      # per base: its insert ...
        b = np.arange(int(off[-1]), dtype=np.int64) - off[:-1][rec_of]       # ... and its index inside it
        ch = draw4(key, seq, (b >> 6), O[ins][rec_of], TAG_INS)
        j = b & 63
        w = np.choose(j >> 4, ch)
        code = (w >> (U64(2) * (j & 15).astype(np.uint64))) & U64(3)
        pool = np.frombuffer(b"ATGC", dtype=np.uint8)[code.astype(np.int64)]
    recs = list(zip(P.tolist(), S.tolist(), extra.tolist(), T.tolist(), aux.tolist()))
    return recs, pool.tobytes(), len(kept) == 0
