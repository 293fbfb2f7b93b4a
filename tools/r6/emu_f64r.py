#!/usr/bin/env python3
"""Index algebra of f64r_kernel (csrc/ro_f64reg.hip) emulated on the CPU: the register-resident FP64 transform.

bins N = D x M, M = 16 x 16 x 16 x R3 in {4096, 8192, 16384}, D in {1, 2, 4}; one workgroup of T = M / 16 threads holds
the M complex doubles of sub-row q (bins q + D k) as 16 points per thread.  Restated with numpy and checked BEFORE
anything runs on a GPU:
  1. the twisted radix-16 / radix-R3 butterflies with merged twiddles (eight table entries per radix-16 pass)
     against their definition  out[bitrev(k)] = sum_j x[j] w^j W_R^(j k);
  2. every thread map, LDS address, twiddle-table index and window-table index, end to end against numpy's FFT of the
     windowed row, for every (M, D) the library routes to the kernel;
  3. that a wave only touches its own territory between barrier (d) and barrier (e);
  4. bank conflicts per wave-instruction (ds_write_b64: 4 x 16 lanes over 32 banks of 8 bytes... see lds_conflicts).
The constants here are the ones of csrc/ro_f64reg.hip (tests/test_layout_cpu.py greps them)."""
import sys
import numpy as np


def bitrev(k, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (k & 1)
        k >>= 1
    return r


BR4 = [bitrev(k, 4) for k in range(16)]


class Geometry:
    def __init__(self, M, D):
        assert M in (4096, 8192, 16384) and D in (1, 2, 4, 8)
        self.M, self.D, self.N = M, D, M * D
        self.T = M // 16
        self.R3 = M // 4096
        self.Q = self.T // 16                      # threads per k0 group = 16 R3
        self.ST = self.T + 16 * self.R3            # doubles per k0 territory of a plane
        self.S2 = self.Q + (1 if self.R3 == 1 else 2)   # exchange 2: stride between k1 slots
        self.PLANE = 16 * self.ST                  # doubles
        self.L3 = {1: 0, 2: 1, 4: 2}[self.R3]


def W(n, e):
    return np.exp(-2j * np.pi * (np.asarray(e, dtype=np.float64) % n) / n)


# ---------------------------------------------------------------------------------------------------------------
# tables (host side of the library: ro_f64reg_tables)
# ---------------------------------------------------------------------------------------------------------------
def tw8(base_n, base_e):
    """the eight merged twiddles of a twisted radix-16 pass with twist w = W_base_n^base_e:
    {w^8, w^4, w^2, w^2 W16^2, w, w W16, w W16^2, w W16^3}"""
    e = np.asarray(base_e, dtype=np.int64)
    n = base_n
    # everything on the common denominator 16 n (exact integer exponents)
    def wp(p, c):  # w^p * W16^c
        return W(16 * n, (16 * e * p + c * n) % (16 * n))
    return np.stack([wp(8, 0), wp(4, 0), wp(2, 0), wp(2, 2), wp(1, 0), wp(1, 1), wp(1, 2), wp(1, 3)], axis=-1)


def tables(g):
    D, N = g.D, g.N
    t = {}
    # pass 0: twist W_(16 D)^q, uniform per q
    t["p0"] = np.stack([tw8(16 * D, q) for q in range(D)])                          # [D][8]
    # pass 1: twist W_(256 D)^(q + D k0)
    t["p1"] = np.stack([tw8(256 * D, q + D * np.arange(16)) for q in range(D)])     # [D][16][8]
    # pass 2: twist W_(4096 D)^(q + D K1), K1 = k0 + 16 k1
    t["p2"] = np.stack([tw8(4096 * D, q + D * np.arange(256)) for q in range(D)])   # [D][256][8]
    # pass 3: a' = W_N^(q + D (K1 + 256 g)), {a', a'^2}, indexed [q][K1 * R3 + g]
    if g.R3 > 1:
        K1 = np.arange(256)[:, None]
        gg = np.arange(g.R3)[None, :]
        a = []
        for q in range(D):
            e = (q + D * (K1 + 256 * gg)).reshape(-1)
            a.append(np.stack([W(N, e), W(N, 2 * e)], axis=-1))
        t["p3"] = np.stack(a)                                                       # [D][256 R3][2]
    return t


# ---------------------------------------------------------------------------------------------------------------
# butterflies
# ---------------------------------------------------------------------------------------------------------------
def bfly(a, b, t, rot):
    t = t * (-1j) ** rot
    return a + t * b, a - t * b


def twisted16(v, tw):
    """v [..., 16], tw [..., 8]; in place, result k at position bitrev4(k)"""
    v = v.copy()
    for lvl in range(4):
        L = 16 >> lvl
        h = L // 2
        for beta in range(1 << lvl):
            kappa = bitrev(beta, lvl)
            e = kappa * h                              # exponent of W16
            idx = {0: 0, 1: 1, 2: 2 + (e % 4) // 2, 3: 4 + e % 4}[lvl]
            rot = e // 4
            for m in range(h):
                i0, i1 = beta * L + m, beta * L + m + h
                v[..., i0], v[..., i1] = bfly(v[..., i0], v[..., i1], tw[..., idx], rot)
    return v


def twisted_small(v, th, R):
    """R in {2, 4}: v [..., R], th = twist; result k at bitrev(k)"""
    v = v.copy()
    if R == 2:
        v[..., 0], v[..., 1] = bfly(v[..., 0], v[..., 1], th, 0)
    else:
        t2 = th * th
        v[..., 0], v[..., 2] = bfly(v[..., 0], v[..., 2], t2, 0)
        v[..., 1], v[..., 3] = bfly(v[..., 1], v[..., 3], t2, 0)
        v[..., 0], v[..., 1] = bfly(v[..., 0], v[..., 1], th, 0)
        v[..., 2], v[..., 3] = bfly(v[..., 2], v[..., 3], th, 1)
    return v


def check_butterflies():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    for n, e in ((256, 37), (4096, 1234), (16, 0), (32, 1)):
        w = W(n, e)
        got = twisted16(x[None, :], tw8(n, np.array([e])))[0]
        want = np.array([sum(x[j] * w ** j * W(16, j * k) for j in range(16)) for k in range(16)])
        assert np.allclose(got[BR4], want, atol=1e-12), (n, e)
    for R in (2, 4):
        y = x[:R]
        th = W(16384, 777)
        got = twisted_small(y[None, :], np.array([th]), R)[0]
        want = np.array([sum(y[j] * th ** j * W(R, j * k) for j in range(R)) for k in range(R)])
        assert np.allclose(got[[bitrev(k, R.bit_length() - 1) for k in range(R)]], want, atol=1e-12)
    print("butterflies ok")


# ---------------------------------------------------------------------------------------------------------------
# LDS model
# ---------------------------------------------------------------------------------------------------------------
class LDS:
    def __init__(self, g):
        self.g = g
        self.mem = np.full(g.PLANE, np.nan)            # doubles
        self.conf = {}
        self.territory_check = False

    def _conf(self, name, addr_bytes, width):
        """worst number of LDS cycles per wave-instruction, relative to the conflict-free count, for one access
        pattern.  ds_read_b64: 2 groups of 32 lanes over 64 banks of 4 B; ds_write_b64: 4 groups of 16 lanes over
        32 banks (MI355X_MICROARCH.md, LDS); ds_read_b32 / ds_write_b32: 2 x 32 over 32 banks."""
        a = np.asarray(addr_bytes).reshape(-1, 64)
        worst = 1
        if name.endswith("r64"):
            groups, banks = 2, 64
        elif name.endswith("w64"):
            groups, banks = 4, 32
        else:
            groups, banks = 2, 32
        per = 64 // groups
        for wv in a:
            for gi in range(groups):
                ad = wv[gi * per:(gi + 1) * per]
                dw = np.unique(ad // 4)                 # identical addresses broadcast
                if width == 8:
                    dw = np.unique(np.concatenate([dw, dw + 1]))
                cnt = np.bincount(dw % banks, minlength=banks).max()
                worst = max(worst, cnt)
        self.conf[name] = max(self.conf.get(name, 1), worst)

    def check_territory(self, addr_d, k0_of_lane):
        if self.territory_check:
            assert np.all(addr_d // self.g.ST == k0_of_lane), "a wave left its territory"

    def write(self, name, addr_d, val, k0=None):
        addr_d = np.asarray(addr_d)
        assert addr_d.max() < self.g.PLANE and addr_d.min() >= 0
        assert len(np.unique(addr_d)) == addr_d.size, name + ": two lanes write one cell"
        if k0 is not None:
            self.check_territory(addr_d, k0)
        self._conf(name + ".w64", addr_d * 8, 8)
        self.mem[addr_d] = val

    def read(self, name, addr_d, k0=None):
        addr_d = np.asarray(addr_d)
        if k0 is not None:
            self.check_territory(addr_d, k0)
        self._conf(name + ".r64", addr_d * 8, 8)
        out = self.mem[addr_d]
        assert not np.isnan(out).any(), name + ": read of a cell nobody wrote"
        return out


def image_cell(g, k0, s, u):
    """float index of magnitude (k0, slot s, lane-in-group u) inside the plane (floats = 2 x doubles); wave k0-group
    writes only into its own territory"""
    if g.R3 == 1:
        c = k0 >> 2                                  # conflict-free for the read-out; the write is 2-way (free for b32)
        return k0 * 2 * g.ST + (s ^ (c & 1)) * 16 + ((u + 8 * (c >> 1)) & 15)
    return k0 * 2 * g.ST + s * g.Q + ((u + 2 * k0) & (g.Q - 1))


def emulate(M, D, q, x, w, verbose=False):
    """one sub-row: x [N] complex samples (already with gain), w [N] window -> magnitudes of bins q + D k, k < M, and the
    global column each lane stores to"""
    g = Geometry(M, D)
    T, Q, R3, ST, N = g.T, g.Q, g.R3, g.ST, g.N
    tb = tables(g)
    t = np.arange(T)
    lds = LDS(g)

    # ---- pass 0: thread t = n1, slot n0: folded sample i = t + T n0
    v = np.zeros((T, 16), dtype=np.complex128)
    for n0 in range(16):
        i = t + T * n0
        acc = np.zeros(T, dtype=np.complex128)
        for r in range(D):
            acc = acc + W(D, r * q) * w[i + M * r] * x[i + M * r]
        v[:, n0] = acc
    v = twisted16(v, np.broadcast_to(tb["p0"][q], (T, 8)))

    # ---- exchange 1 (workgroup-wide), plane by plane
    k0 = t // Q
    n2 = t % Q
    xin = np.zeros((T, 16), dtype=np.complex128)
    for part in ("real", "imag"):
        lds.mem[:] = np.nan
        for kk in range(16):
            lds.write("x1", kk * ST + t, getattr(v[:, BR4[kk]], part))
        got = np.zeros((T, 16))
        for j1 in range(16):
            got[:, j1] = lds.read("x1", k0 * ST + n2 + Q * j1)
        if part == "real":
            xin.real = got
        else:
            xin.imag = got
    lds.territory_check = True

    # ---- pass 1: thread (k0, n2), twist W_(256 D)^(q + D k0)
    v = twisted16(xin, tb["p1"][q][k0])

    # ---- exchange 2 (inside the k0 group): slot k1 -> cell k1 S2 + n2;  thread (k0, k1, n3) at u = n3 16 + k1 reads
    # n2 = n3 + R3 j2
    u = t % Q
    k1 = u % 16
    n3 = u // 16
    xin = np.zeros((T, 16), dtype=np.complex128)
    for part in ("real", "imag"):
        lds.mem[:] = np.nan
        for kk in range(16):
            lds.write("x2", k0 * ST + kk * g.S2 + n2, getattr(v[:, BR4[kk]], part), k0)
        got = np.zeros((T, 16))
        for j2 in range(16):
            got[:, j2] = lds.read("x2", k0 * ST + k1 * g.S2 + n3 + R3 * j2, k0)
        if part == "real":
            xin.real = got
        else:
            xin.imag = got

    # ---- pass 2: thread (k0, k1, n3), twist W_(4096 D)^(q + D K1)
    K1 = k0 + 16 * k1
    v = twisted16(xin, tb["p2"][q][K1])

    # ---- exchange 3 by lane swaps + pass 3
    lane = t & 63
    if R3 > 1:
        def swap(v, bit, pos_bit):
            """v_permlane32_swap (bit 5) / v_permlane16_swap (bit 4) of positions (p, p + 2^pos_bit): a 2 x 2 transposition
            between that lane bit and that position bit: new[lane(b), pos(a)] = old[lane(b = a), pos(a = b)]"""
            out = v.copy()
            tt = np.arange(T)
            lb = (lane >> bit) & 1
            for p in range(16):
                a = (p >> pos_bit) & 1
                src_t = (tt & ~(1 << bit)) | (a << bit)              # same thread but lane bit = a
                src_p = (p & ~(1 << pos_bit))                        # position bit = the reader's lane bit
                out[:, p] = v[src_t, src_p | (lb << pos_bit)]
            return out
        if R3 == 4:
            v = swap(v, 5, 3)
            v = swap(v, 4, 2)
            b5, b4 = (lane >> 5) & 1, (lane >> 4) & 1
            gp = b5 + 2 * b4
            def P(n, i):
                return 8 * (n >> 1) + 4 * (n & 1) + 2 * (i & 1) + (i >> 1)
            per = 4
        else:
            v = swap(v, 4, 3)
            gp = (lane >> 4) & 1
            def P(n, i):
                return 8 * n + bitrev(i, 3)
            per = 8
        a1 = tb["p3"][q][K1 * R3 + gp, 0]
        a2 = tb["p3"][q][K1 * R3 + gp, 1]
        out = np.zeros((T, 16), dtype=np.complex128)
        sbin = [0] * 16
        for i in range(per):
            th = a1 * W(16, i)
            assert np.allclose(a2 * W(8, i), th * th, atol=1e-13)
            idx = [P(n, i) for n in range(R3)]
            blk = twisted_small(v[:, idx], th, R3)
            for n in range(R3):                                       # result k3 at the position of n = bitrev(k3)
                out[:, idx[n]] = blk[:, n]
                sbin[idx[n]] = 256 * R3 * i + 4096 * bitrev(n, g.L3)
        v = out
        lane_bin = k0 + 16 * k1 + 256 * gp
    else:
        sbin = [256 * BR4[s] for s in range(16)]
        lane_bin = k0 + 16 * k1

    # ---- magnitude image: float cells inside the own territory, then the read-out
    img = np.full(2 * g.PLANE, np.nan)
    cell_bin = np.full(2 * g.PLANE, -1)
    for s in range(16):
        c = image_cell(g, k0, s, u)
        assert np.all(c // (2 * ST) == k0)
        assert len(np.unique(c)) == T
        lds._conf("img.w32", c * 4, 4)
        img[c] = np.abs(v[:, s])
        cell_bin[c] = lane_bin + sbin[s]
    wv = t >> 6
    mags = np.full(M, np.nan)
    cols = np.full(M, -1)
    if D == 1:
        # 16-byte stores: lane = c + 4 ul, c = k0 >> 2, ul = u & 15; 4 stores per thread; the remaining bits of (u, s)
        # from (wave, it): u = ul + 16 uh, code = wave + waves * it = uh + R3 * s
        rc, ul = lane & 3, lane >> 2
        waves = T // 64
        for it in range(4):
            code = wv + waves * it
            uh, s = code % R3, code // R3
            ru = ul + 16 * uh
            quad = []
            for e in range(4):
                c = image_cell(g, 4 * rc + e, s, ru)
                lds._conf("imgw.r32", c * 4, 4)
                quad.append(c)
            kb = np.stack([cell_bin[c] for c in quad], axis=1)             # [T][4] bins of a lane's 16-byte store
            assert np.all(np.diff(kb, axis=1) == 1) and np.all(kb[:, 0] % 4 == 0)
            # a wave's store is one contiguous run of 256 bins
            first = kb[:, 0].reshape(-1, 64)
            assert np.all(np.diff(first, axis=1) == 4)
            for e in range(4):
                mags[kb[:, e]] = img[quad[e]]
                cols[kb[:, e]] = (kb[:, e] + N // 2) % N
    else:
        # 4-byte stores (a sub-row owns every D-th column): lane l of wave wv, iteration it: k0 = l & 15,
        # u = (l >> 4) | (wv << 2), s = it
        for it in range(16):
            rk0 = lane & 15
            ru = (lane >> 4) | (wv << 2)
            c = image_cell(g, rk0, it, ru)
            lds._conf("img.r32", c * 4, 4)
            kbin = cell_bin[c]
            assert (kbin >= 0).all()
            mags[kbin] = img[c]
            cols[kbin] = (q + D * kbin + N // 2) % N
            kb = kbin.reshape(-1, 64)
            assert np.all(np.diff(kb, axis=1) == 1)                         # 64 consecutive sub-row bins per wave
    assert not np.isnan(mags).any()
    return mags, cols, lds.conf


def twisted16_levels(v, tw, levels):
    """the first `levels` levels of twisted16: a twisted radix-2^levels transform over the TOP bits of the position, one
    per value of the low bits; result kd at top bits = bitrev(kd)"""
    v = v.copy()
    for lvl in range(levels):
        L = 16 >> lvl
        h = L // 2
        for beta in range(1 << lvl):
            e = bitrev(beta, lvl) * h
            idx = {0: 0, 1: 1, 2: 2 + (e % 4) // 2, 3: 4 + e % 4}[lvl]
            rot = e // 4
            for m in range(h):
                i0, i1 = beta * L + m, beta * L + m + h
                v[..., i0], v[..., i1] = bfly(v[..., i0], v[..., i1], tw[..., idx], rot)
    return v


def emulate_tile(N, x, w):
    """bins N = 256 R < 4096: B = 4096 / N = 16 / R ROWS in the M = 4096 workgroup (f64r_kernel<12, 1, ., ., LOGB>).  The tile
    index i = n2 + 16 j1 + 256 n0 carries the row in the LOW bits of n2 = b + B m, and sample i' = m + R j1 + 16 R n0 of
    row b: passes 0 and 1 are the 4096-point kernel's own, with its own tables, and the row's last digit m is what the first
    log2(R) levels of pass 2 transform -- again with the 4096-point table: W_4096^(K1 B m) = W_N^(K1 m).
    x [B][N] windowed-to-be rows, w [N] -> magnitudes [B][N] by bin, columns [B][N]"""
    g = Geometry(4096, 1)
    T, Q, ST = g.T, g.Q, g.ST
    B = 4096 // N
    R = 16 // B
    r = R.bit_length() - 1
    tb = tables(g)
    t = np.arange(T)
    lds = LDS(g)
    # ---- pass 0: thread t = (b, tt), slot n0: sample tt + (N / 16) n0 of row b
    b = t % B
    tt = (t % 16) // B + R * (t // 16)
    assert sorted(zip(b.tolist(), tt.tolist())) == [(bb, q) for bb in range(B) for q in range(N // 16)]
    v = np.zeros((T, 16), dtype=np.complex128)
    for n0 in range(16):
        i = tt + (N // 16) * n0
        v[:, n0] = w[i] * x[b, i]
    # a wave's load of one slot: per row a run of consecutive samples
    for wv in range(T // 64):
        ln = slice(64 * wv, 64 * wv + 64)
        for bb in range(B):
            run_ = np.sort(tt[ln][b[ln] == bb])
            assert np.all(np.diff(run_) == 1) and run_.size == 64 // B
    v = twisted16(v, np.broadcast_to(tb["p0"][0], (T, 8)))
    k0 = t // Q
    n2 = t % Q
    xin = np.zeros((T, 16), dtype=np.complex128)
    for part in ("real", "imag"):
        lds.mem[:] = np.nan
        for kk in range(16):
            lds.write("x1", kk * ST + t, getattr(v[:, BR4[kk]], part))
        got = np.zeros((T, 16))
        for j1 in range(16):
            got[:, j1] = lds.read("x1", k0 * ST + n2 + Q * j1)
        if part == "real":
            xin.real = got
        else:
            xin.imag = got
    lds.territory_check = True
    v = twisted16(xin, tb["p1"][0][k0])
    u = t % Q
    k1 = u % 16
    xin = np.zeros((T, 16), dtype=np.complex128)
    for part in ("real", "imag"):
        lds.mem[:] = np.nan
        for kk in range(16):
            lds.write("x2", k0 * ST + kk * g.S2 + n2, getattr(v[:, BR4[kk]], part), k0)
        got = np.zeros((T, 16))
        for j2 in range(16):
            got[:, j2] = lds.read("x2", k0 * ST + k1 * g.S2 + j2, k0)
        if part == "real":
            xin.real = got
        else:
            xin.imag = got
    # ---- pass 2, its first log2(R) levels only: position p = b + B m
    K1 = k0 + 16 * k1
    v = twisted16_levels(xin, tb["p2"][0][K1], r)
    # slot s: row s % B, k2 = bitrev_r(s / B); bin k0 + 16 k1 + 256 k2
    img = np.full(2 * g.PLANE, np.nan)
    cell_bin = np.full(2 * g.PLANE, -1)
    cell_row = np.full(2 * g.PLANE, -1)
    for s_ in range(16):
        c = image_cell(g, k0, s_, u)
        lds._conf("img.w32", c * 4, 4)
        img[c] = np.abs(v[:, s_])
        cell_bin[c] = k0 + 16 * k1 + 256 * bitrev(s_ // B, r)
        cell_row[c] = s_ % B
    lane = t & 63
    wv = t >> 6
    mags = np.full((B, N), np.nan)
    cols = np.full((B, N), -1)
    rc, ul = lane & 3, lane >> 2
    for it in range(4):
        s_ = wv + 4 * it                                                  # (wave-uniform: the row of a store is, too)
        quad = [image_cell(g, 4 * rc + e, s_, ul) for e in range(4)]
        for c in quad:
            lds._conf("imgw.r32", c * 4, 4)
        kb = np.stack([cell_bin[c] for c in quad], axis=1)
        rb = np.stack([cell_row[c] for c in quad], axis=1)
        assert np.all(rb == (s_ % B)[:, None]) and np.all(np.diff(kb, axis=1) == 1) and np.all(kb[:, 0] % 4 == 0)
        first = kb[:, 0].reshape(-1, 64)
        assert np.all(np.diff(first, axis=1) == 4)
        assert np.all(kb[:, 0] == 4 * rc + 16 * ul + 256 * np.array([bitrev(int(q) // B, r) for q in s_]))
        for e in range(4):
            mags[rb[:, e], kb[:, e]] = img[quad[e]]
            cols[rb[:, e], kb[:, e]] = (kb[:, e] + N // 2) % N
    assert not np.isnan(mags).any()
    return mags, cols, lds.conf


def run_tile(N, seed=0):
    B = 4096 // N
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))
    w = rng.random(N)
    mags, cols, conf = emulate_tile(N, x, w)
    err = 0.0
    for b in range(B):
        want = np.abs(np.fft.fft(x[b] * w))
        row = np.full(N, np.nan)
        row[cols[b]] = mags[b]
        err = max(err, np.abs(row - np.roll(want, N // 2)).max() / want.max())
    assert err < 1e-12, (N, err)
    return err, conf


def run(M, D, seed=0):
    g = Geometry(M, D)
    N = g.N
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    w = rng.random(N)
    want = np.abs(np.fft.fft(x * w))
    row = np.full(N, np.nan)
    conf = {}
    for q in range(D):
        mags, cols, c = emulate(M, D, q, x, w)
        row[cols] = mags
        for k, val in c.items():
            conf[k] = max(conf.get(k, 1), val)
    want_shift = np.roll(want, N // 2)                  # column (k + N/2) mod N = |X[k]|
    err = np.abs(row - want_shift).max() / want.max()
    assert err < 1e-12, (M, D, err)
    return err, conf


def main():
    check_butterflies()
    worst_conf = {}
    for M, D in ((4096, 1), (8192, 1), (16384, 1), (16384, 2), (16384, 4)):     # what f64r::plan() routes
        if True:
            err, conf = run(M, D, seed=M + D)
            print("M = %5d  D = %d  (bins %6d): max err / row max %.2e   LDS cycles vs conflict-free: %s"
                  % (M, D, M * D, err, "  ".join("%s %dx" % (k, v) for k, v in sorted(conf.items()))))
            for k, v in conf.items():
                worst_conf[k] = max(worst_conf.get(k, 1), v)
    for N in (256, 512, 1024, 2048):                                              # rows batched into the M = 4096 workgroup
        err, conf = run_tile(N, seed=N)
        print("bins %4d, %2d rows per workgroup: max err / row max %.2e   LDS cycles vs conflict-free: %s"
              % (N, 4096 // N, err, "  ".join("%s %dx" % (k, v) for k, v in sorted(conf.items()))))
        for k, v in conf.items():
            worst_conf[k] = max(worst_conf.get(k, 1), v)
    # ds_read_b64 / ds_write_b64 count 2 dwords per lane: "1x" means one pass per lane group
    # (a 2-way conflict on ds_write_b32 costs nothing: the store's data transfer takes twice its LDS-array cycles)
    bad = {k: v for k, v in worst_conf.items() if v > (2 if k == "img.w32.w32" or k == "img.w32" else 1)}
    if bad:
        print("bank conflicts:", bad)
        return 1
    print("all maps ok, every LDS access conflict-free")
    return 0


if __name__ == "__main__":
    sys.exit(main())
