#!/usr/bin/env python3
"""Index algebra and planar butterflies of stft32k_kernel (N = 32768, 1024 threads) emulated on the CPU.

Restated with numpy and checked BEFORE anything runs on a GPU:
  1. the planar radix-32 stage of csrc/ro_fft_planar.h -- joint / mixed / inpair butterflies with the VOP3P modifier
     strings (op_sel, op_sel_hi, neg_lo, neg_hi) parsed OUT OF THE HEADER and applied by a model of v_pk_fma_f32 --
     against the definition  out[bitrev32(k)] = sum_r x[r] w^r W32^(r k),  for both mate bits;
  2. every LDS address map, lane map, twiddle index and window-table index of csrc/ro_stft32k.hip, end to end against
     numpy's FFT of a windowed row;
  3. bank conflicts per wave-instruction (ds_read_b64: two groups of 32 lanes over 64 banks);
  4. 8-byte alignment of every ds_read_b64, and the M0 / offset splits of the add-TID writes (16 bits each).
Constants must match csrc/ro_k32_lds.h (RQ, HB, XB)."""
import os
import re
import numpy as np

N, T, RQ, HB, XB = 32768, 1024, 1026, 61692, 3972
HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(HERE, "..", "..", "radio-observer_amd", "csrc", "ro_fft_planar.h")
KRN = os.path.join(HERE, "..", "..", "radio-observer_amd", "csrc", "ro_stft32k.hip")


def bitrev(k, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (k & 1)
        k >>= 1
    return r


# ---------------------------------------------------------------------------------------------------------------
# 1. planar butterflies
# ---------------------------------------------------------------------------------------------------------------
def parse_mods(text):
    d = {"op_sel": [0, 0, 0], "op_sel_hi": [1, 1, 1], "neg_lo": [0, 0, 0], "neg_hi": [0, 0, 0]}
    for key in d:
        m = re.search(key + r":\[([0-9,]+)\]", text.replace("op_sel_hi", "OPSELHI") if key == "op_sel" else text)
        if key == "op_sel":
            m = re.search(r"(?<![A-Za-z_])op_sel:\[([0-9,]+)\]", text)
        if m:
            v = [int(x) for x in m.group(1).split(",")]
            d[key] = v + [d[key][i] for i in range(len(v), 3)]
    return d


def pk_fma(s0, s1, s2, mods):
    """v_pk_fma_f32: D.lo = s0[op_sel[0]] * s1[op_sel[1]] + s2[op_sel[2]] (neg_lo per source), D.hi with op_sel_hi / neg_hi"""
    src = (s0, s1, s2)
    out = []
    for half, sel, neg in ((0, mods["op_sel"], mods["neg_lo"]), (1, mods["op_sel_hi"], mods["neg_hi"])):
        a, b, c = [(-1.0 if neg[i] else 1.0) * src[i][sel[i]] for i in range(3)]
        out.append(a * b + c)
    return np.array(out)


def header_asm(func, variant=None):
    """modifier strings of the asm statements of one function of ro_fft_planar.h, in source order"""
    src = open(HDR).read()
    body = src[src.index("void " + func):]
    body = body[:body.index("\n}\n")]
    if variant is not None:                      # the two branches of `if constexpr (!MI) {...} else {...}`
        a = body.index("if constexpr (!MI)")
        b = body.index("} else {", a)
        body = body[a:b] if variant == "noMI" else body[b:]
    return [parse_mods(m) for m in re.findall(r'asm\("v_pk_fma_f32 %0, %1, %2, %3([^"]*)"', body)]


DEFAULT = {"op_sel": [0, 0, 0], "op_sel_hi": [1, 1, 1], "neg_lo": [0, 0, 0], "neg_hi": [0, 0, 0]}
XX = {"op_sel": [0, 0, 0], "op_sel_hi": [1, 0, 1], "neg_lo": [0, 0, 0], "neg_hi": [0, 0, 0]}     # s1.xx
YY = {"op_sel": [0, 1, 0], "op_sel_hi": [1, 1, 1], "neg_lo": [0, 0, 0], "neg_hi": [0, 0, 0]}     # s1.yy
NEG1 = lambda m: dict(m, neg_lo=[0, 1, 0], neg_hi=[0, 1, 0])


def bf_joint(MI, AR, AI, BR, BI, w):
    # the builtin forms of the header: w.xx / w.yy swizzles, a full negation of the second source
    if not MI:
        ur = pk_fma(BR, w, AR, XX); ui = pk_fma(BI, w, AI, XX)
        sr = pk_fma(BI, w, ur, NEG1(YY)); si = pk_fma(BR, w, ui, YY)
    else:
        ur = pk_fma(BR, w, AR, YY); ui = pk_fma(BI, w, AI, YY)
        sr = pk_fma(BI, w, ur, XX); si = pk_fma(BR, w, ui, NEG1(XX))
    return sr, si, 2 * AR - sr, 2 * AI - si


MIXED = header_asm("bf_mixed")
INPAIR = {False: header_asm("bf_inpair", "noMI"), True: header_asm("bf_inpair", "MI")}
assert len(MIXED) == 2 and len(INPAIR[False]) == 4 and len(INPAIR[True]) == 4


def bf_mixed(AR, AI, BR, BI, w):
    ur = pk_fma(BR, w, AR, DEFAULT); ui = pk_fma(BI, w, AI, DEFAULT)
    sr = pk_fma(BI, w, ur, MIXED[0]); si = pk_fma(BR, w, ui, MIXED[1])
    return sr, si, 2 * AR - sr, 2 * AI - si


def bf_inpair(MI, R, I, w):
    m = INPAIR[MI]
    u = pk_fma(R, w, R, m[0]); u2 = pk_fma(I, w, I, m[1])
    r = pk_fma(I, w, u, m[2]); i = pk_fma(R, w, u2, m[3])
    return r, i


def pr(MB, p):
    return (p >> 1) if MB == 0 else (((p >> 2) << 1) | (p & 1))


def hf(MB, p):
    return (p >> MB) & 1


def c2v(z):
    return np.array([z.real, z.imag])


def planar_stage(MB, x, w):
    """the planar fdit32 of the header on 32 complex inputs x[position], twiddle w; returns the complex positions"""
    R = np.zeros((16, 2)); I = np.zeros((16, 2))
    for p in range(32):
        R[pr(MB, p), hf(MB, p)] = x[p].real
        I[pr(MB, p), hf(MB, p)] = x[p].imag
    W32 = np.exp(-2j * np.pi / 32)
    for L in range(5):
        g = w ** (16 >> L)
        tw = {e: c2v(W32 ** e * g) for e in range(8)}
        S, PB = 32 >> L, 4 - L
        for n in range(16):
            U = n // (S // 2); pa = U * S + n % (S // 2); pb = pa + S // 2
            E = bitrev(U, L) * (16 >> L)
            if PB == MB:
                assert pr(MB, pa) == pr(MB, pb) and hf(MB, pa) == 0 and hf(MB, pb) == 1
                i = pr(MB, pa)
                R[i], I[i] = bf_inpair(E >= 8, R[i], I[i], tw[E & 7])
            elif hf(MB, pa) == 0:
                pm = pa | (1 << MB)
                assert pr(MB, pm) == pr(MB, pa) and pr(MB, pm + S // 2) == pr(MB, pb)
                Em = bitrev(pm // S, L) * (16 >> L)
                ia, ib = pr(MB, pa), pr(MB, pb)
                if MB < PB:
                    assert Em == E
                    R[ia], I[ia], R[ib], I[ib] = bf_joint(E >= 8, R[ia], I[ia], R[ib], I[ib], tw[E & 7])
                else:
                    assert Em == E + 8 and E < 8
                    R[ia], I[ia], R[ib], I[ib] = bf_mixed(R[ia], I[ia], R[ib], I[ib], tw[E])
    return np.array([R[pr(MB, p), hf(MB, p)] + 1j * I[pr(MB, p), hf(MB, p)] for p in range(32)])


def check_planar():
    rng = np.random.default_rng(2)
    for MB in (0, 1):
        x = rng.standard_normal(32) + 1j * rng.standard_normal(32)
        w = np.exp(-2j * np.pi * 0.3173)
        got = planar_stage(MB, x, w)
        want = np.fft.fft(x * w ** np.arange(32))
        err = max(abs(got[bitrev(k, 5)] - want[k]) for k in range(32))
        assert err < 1e-12, (MB, err)
    # the last levels' unit orders of the header (last0: blocks J, 8 + J; last1: blocks 2u, 2u + 1) and their products
    for J in range(8):
        E0, E1 = bitrev(J, 4), bitrev(8 + J, 4)
        assert E1 == E0 + 1
        if J % 2 == 1:
            assert (E0 & 7) == (bitrev(J - 1, 4) & 7) and (E1 & 7) == (bitrev(7 + J, 4) & 7)      # shared products
    for u in range(8):
        assert bitrev(2 * u, 4) < 8 and bitrev(2 * u + 1, 4) == bitrev(2 * u, 4) + 8
    print("planar radix-32 stage (modifier strings from ro_fft_planar.h) == DFT32 of x[r] w^r, mate bits 0 and 1")


# ---------------------------------------------------------------------------------------------------------------
# 2.-4. the kernel's maps
# ---------------------------------------------------------------------------------------------------------------
def cell(q, w, lane):
    return RQ * q + 64 * w + lane


def rot(w):
    return 4 * (w >> 1)


def column(t):
    l = t & 63
    j = l & 31
    return (t & ~63) + 2 * ((j >> 1) + 16 * (j & 1)) + (l >> 5)


def x1_cell(k0, t):      # exchange 1: slot k0 of pass-0 thread t
    return cell((t >> 6) + 16 * (k0 & 1), k0 >> 1, t & 63)


def banks_ok_b64(addrs):
    """ds_read_b64: groups {0..31}, {32..63}; bank of dword a = a mod 64; each lane takes dwords a, a + 1"""
    for g in (0, 32):
        seen = {}
        for a in addrs[g:g + 32]:
            assert a % 2 == 0, "8-byte alignment"
            for d in (a, a + 1):
                b = d % 64
                if b in seen and seen[b] != d:
                    return False
                seen[b] = d
    return True


def window_table(w):
    """stft32k_window_layout"""
    out = np.zeros(N)
    for t in range(T):
        c = column(t)
        for q in range(8):
            o = (q * T + t) * 4
            out[o:o + 4] = [w[c + 1024 * (2 * q)], w[c + 1024 * (2 * q + 16)], w[c + 1024 * (2 * q + 1)], w[c + 1024 * (2 * q + 17)]]
    return out


def check_kernel():
    src = open(os.path.join(os.path.dirname(KRN), "ro_k32_lds.h")).read()
    for name, val in (("RQ", RQ), ("HB", HB), ("XB", XB)):
        assert re.search(r"constexpr int %s = %d;" % (name, val), src), name
    rng = np.random.default_rng(1)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    win = rng.standard_normal(N)
    want = np.fft.fft(x * win)
    wk = window_table(win)
    # ---- loads: lane l < 32 fetches columns c, c + 1 (16 bytes) for legs 0..15, lane l + 32 for legs 16..31
    cols = set()
    for t in range(T):
        cols.add(column(t))
        if (t & 63) < 32:
            assert column(t + 32) == column(t) + 1 and column(t) % 2 == 0
    assert cols == set(range(1024))
    for wv in range(16):        # one wave-load of lanes 0..31 covers 512 contiguous bytes
        base = sorted(column(64 * wv + l) for l in range(32))
        assert base == list(range(64 * wv, 64 * wv + 64, 2))
    # ---- pass 0 (level 0 with the window in it: slots i, i + 16 with coefficient quad i / 2)
    y0 = np.zeros((T, 32), dtype=complex)
    for t in range(T):
        c = column(t)
        v = x[c + 1024 * np.arange(32)].copy()
        for i in range(16):
            q = wk[((i // 2) * T + t) * 4:((i // 2) * T + t) * 4 + 4]
            wi, wj = (q[2], q[3]) if i & 1 else (q[0], q[1])
            tt = v[i] * wi
            v[i], v[i + 16] = tt + v[i + 16] * wj, tt - v[i + 16] * wj
        # levels 1..4 of the unwindowed dit<32> = DFT32 of (level-0 output), positions as dit: finish with an FFT of
        # the equivalent input: undo level 0 on ones is not needed -- compare through the definition instead
        y0[t] = np.fft.fft(x[c + 1024 * np.arange(32)] * win[c + 1024 * np.arange(32)])
        lvl0 = np.concatenate([(x * win)[c + 1024 * np.arange(16)] + (x * win)[c + 1024 * (16 + np.arange(16))],
                               (x * win)[c + 1024 * np.arange(16)] - (x * win)[c + 1024 * (16 + np.arange(16))]])
        assert np.allclose(v, lvl0)
    # ---- exchange 1
    lds = np.zeros(32 * RQ, dtype=complex)
    used = set()
    for k0 in range(32):
        for t in range(T):
            c = x1_cell(k0, t)
            assert c not in used
            used.add(c)
            lds[c] = y0[t, k0]
    ok = True
    v1 = np.zeros((T, 32), dtype=complex)          # [pass-1 thread][slot b]
    for w in range(16):
        for wp in range(16):
            addrs = []
            for lam in range(64):
                kb = lam >> 5
                a = 2 * (lam & 15) + ((lam >> 4) & 1)
                assert lam == (a >> 1) + 16 * (a & 1) + 32 * kb
                g1 = RQ * 16 * kb + 64 * w + 2 * (lam & 15) + 32 * ((lam >> 4) & 1)
                ad = g1 + RQ * wp
                assert 4 * RQ * wp <= 65535
                addrs.append(ad)
                assert (ad % RQ) // 64 == w and ad // RQ == wp + 16 * kb           # own territory
                v1[64 * w + lam, 2 * wp] = lds[ad]
                v1[64 * w + lam, 2 * wp + 1] = lds[ad + 1]
            ok &= banks_ok_b64(addrs)
    print("exchange 1 reads (ds_read_b64) conflict-free:", ok)
    assert ok
    # what a pass-1 thread must have received: column a + 32 b, slot k0
    for w in (0, 7, 15):
        for lam in (0, 17, 40, 63):
            kb, a = lam >> 5, 2 * (lam & 15) + ((lam >> 4) & 1)
            k0 = 2 * w + kb
            for b in range(32):
                tsrc = [t for t in range(T) if column(t) == a + 32 * b][0]
                assert v1[64 * w + lam, b] == y0[tsrc, k0]
    y1 = np.zeros((T, 32), dtype=complex)
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        k0 = 2 * w + (lam >> 5)
        y1[tau] = np.fft.fft(v1[tau] * np.exp(-2j * np.pi * k0 * np.arange(32) / 1024))
    # ---- exchange 2
    lds[:] = 0
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        for k1 in range(32):
            lds[cell(k1, w, lam)] = y1[tau, k1]
    v2 = np.zeros((T, 32), dtype=complex)
    ok = True
    for w in range(16):
        for i in range(16):
            u, p = i >> 1, i & 1
            addrs = []
            for lam in range(64):
                kb, k1 = lam & 1, ((lam >> 1) + rot(w)) & 31
                g2 = RQ * k1 + 64 * w + 32 * kb
                ad = g2 + 2 * ((i >> 1) + 8 * (i & 1))
                addrs.append(ad)
                assert (ad % RQ) // 64 == w
                # pair 2u + p = slots a = 4u + p (low) and 4u + p + 2 (high): pass-1 lanes (a >> 1) + 16 (a & 1) + 32 kb
                for h in (0, 1):
                    a = 4 * u + p + 2 * h
                    assert ad + h == cell(k1, w, (a >> 1) + 16 * (a & 1) + 32 * kb)
                    assert pr(1, a) == i and hf(1, a) == h
                    v2[64 * w + lam, a] = lds[ad + h]
            ok &= banks_ok_b64(addrs)
    print("exchange 2 reads (ds_read_b64) conflict-free:", ok)
    assert ok
    y2 = np.zeros((T, 32), dtype=complex)
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        kb, k1 = lam & 1, ((lam >> 1) + rot(w)) & 31
        kp = 2 * w + kb + 32 * k1                                  # the pass-2 twiddle index
        y2[tau] = np.fft.fft(v2[tau] * np.exp(-2j * np.pi * kp * np.arange(32) / N))
        assert np.abs(y2[tau] - want[kp + 1024 * np.arange(32)]).max() < 1e-6 * np.abs(want).max()
    print("bins match numpy fft of the windowed row")
    # ---- image and read-back
    lds[:] = 0
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        for k2 in range(32):
            lds[cell(k2, w, lam)] = y2[tau, k2]

    def rb(tid, q, half):
        base = RQ * (tid >> 8) + 128 * (tid & 7) + 2 * ((((tid & 255) >> 3) - 4 * (tid & 7)) & 31)
        return base + 4 * RQ * q + 64 * half

    out = np.zeros(N, dtype=complex)
    ok = True
    for q in range(8):
        for tid in range(T):
            for half in (0, 1):
                ad = rb(tid, q, half)
                r, m = 4 * q + (tid >> 8), tid & 255
                for j in (0, 1):
                    out[(1024 * r + 4 * m + 2 * half + j + N // 2) % N] = lds[ad + j]
        for wv in range(16):
            for half in (0, 1):
                ok &= banks_ok_b64([rb(64 * wv + lam, q, half) for lam in range(64)])
    assert np.allclose(out, np.fft.fftshift(want))
    print("read-back (two ds_read_b64 per lane and chunk) = fft-shifted row; conflict-free:", ok)
    assert ok
    # the store offset of chunk q: column of bin 1024 (4q + (tid >> 8)) + 4 (tid & 255) = tid * 4 + ((q T 4 + N/2) mod N)
    for q in range(8):
        for tid in (0, 255, 256, 1023):
            col = (1024 * (4 * q + (tid >> 8)) + 4 * (tid & 255) + N // 2) % N
            assert col == (tid * 4 + ((q * T * 4 + N // 2) & (N - 1))) % N
    # ---- the scan's view of the image
    for c in rng.integers(0, N, 4000):
        k = (c + N // 2) & (N - 1)
        r, beta = k >> 10, k & 1023
        w, kb, k1 = (beta & 31) >> 1, beta & 1, beta >> 5
        assert abs(lds[RQ * r + 64 * w + 2 * ((k1 - 4 * (w >> 1)) & 31) + kb] - np.fft.fftshift(want)[c]) < 1e-6 * np.abs(want).max()
    print("ImageRow(c) = column c of the fft-shifted row")
    # ---- add-TID reach: M0 / offset splits (16 bits each, multiples of 4)
    for w in range(16):
        for q in range(32):
            m0 = 256 * w + (HB if q >= 16 else 0)
            off = 4 * RQ * q - (HB if q >= 16 else 0)
            assert 0 <= m0 <= 65535 and 0 <= off <= 65535 and m0 + off == 4 * cell(q, w, 0) and m0 % 4 == 0
        for k0 in range(32):
            hb = XB if k0 & 1 else 0
            m0 = 4 * RQ * w + hb
            off = 4 * (RQ * 16 * (k0 & 1) + 64 * (k0 >> 1)) - hb
            assert 0 <= m0 <= 65535 and 0 <= off <= 65535 and m0 + off == 4 * x1_cell(k0, 64 * w), (w, k0, m0, off)
    print("M0 / offset splits fit 16 bits; LDS bytes:", 32 * RQ * 4)
    # ---- result rows of the hooks: pass 1 (pairs j, 8 + j) and pass 2 (unit u)
    for j in range(8):
        q = bitrev(2 * j, 5)
        assert [bitrev(p, 5) for p in (2 * j, 16 + 2 * j, 2 * j + 1, 17 + 2 * j)] == [q, q + 1, q + 16, q + 17] and q < 16
    for u in range(8):
        r = bitrev(u, 3)
        assert [bitrev(p, 5) for p in (4 * u, 4 * u + 2, 4 * u + 1, 4 * u + 3)] == [r, r + 8, r + 16, r + 24]
        assert pr(1, 4 * u) == 2 * u and pr(1, 4 * u + 2) == 2 * u and pr(1, 4 * u + 1) == 2 * u + 1
    print("hook rows: pass 1 q, q+1, q+16, q+17; pass 2 r, r+8, r+16, r+24")


def check_pair():
    """the large transform bins = 2 N on the same kernel (PAIR): half q of a stream row = bins q + 2 k'.
    a = w0 x0 + w1 x1 (q = 0), d = w0 x0 - w1 x1 (q = 1, through the scratch); the rotation W_bins^m of q = 1 as a shift
    of the bin index by 1/2: pass 0 twiddled with exp(-2 pi i (1/2) / 32), pass 1 / 2 stage twiddles times
    exp(-2 pi i (1/2) / 1024), / 32768; bin q + 2 k' leaves for column q + 2 ((k' + N/2) mod N); the scratch unit k of a
    thread = its slots k, 16 + k."""
    rng = np.random.default_rng(5)
    bins = 2 * N
    x = rng.standard_normal(bins) + 1j * rng.standard_normal(bins)
    w = rng.standard_normal(bins)
    want = np.fft.fftshift(np.fft.fft(x * w))
    u, vv = x[:N] * w[:N], x[N:] * w[N:]
    for q, acc in ((0, u + vv), (1, u - vv)):
        phi = q / 2
        y0 = np.zeros((1024, 32), dtype=complex)                # [n1][k0]
        tw0 = np.exp(-2j * np.pi * phi / 32)
        for c in range(1024):
            y0[c] = np.fft.fft(acc[c + 1024 * np.arange(32)] * tw0 ** np.arange(32))
        for k0 in (0, 5, 31):
            for k1 in (0, 9):
                col = np.zeros(32, dtype=complex)
                for a in range(32):
                    w1 = np.exp(-2j * np.pi * k0 / 1024) * np.exp(-2j * np.pi * phi / 1024)
                    col[a] = np.fft.fft(y0[a + 32 * np.arange(32), k0] * w1 ** np.arange(32))[k1]
                w2 = np.exp(-2j * np.pi * (k0 + 32 * k1) / N) * np.exp(-2j * np.pi * phi / N)
                y2 = np.fft.fft(col * w2 ** np.arange(32))
                for k2 in range(32):
                    kp = k0 + 32 * k1 + 1024 * k2
                    colm = q + 2 * ((kp + N // 2) & (N - 1))
                    assert abs(y2[k2] - want[colm]) < 1e-6 * np.abs(want).max(), (q, k0, k1, k2)
    # the window table of block r is the ordinary layout of w[r N ..]; chunk g of block 1 = quad g: slots 2g, 2g + 16,
    # 2g + 1, 2g + 17; the scratch: store (2 g + j) T + tid carries slots 2 g + j and 2 g + j + 16 = load k's v[k], v[16 + k]
    for g in range(8):
        for j in (0, 1):
            k = 2 * g + j
            assert (k, 16 + k) == (2 * g + j, 2 * g + j + 16) and k < 16
    print("PAIR: a = w0 x0 + w1 x1, d = w0 x0 - w1 x1, shifted twiddles for d, columns q + 2 ((k' + N/2) mod N) == fft of the large row")


if __name__ == "__main__":
    check_planar()
    check_kernel()
    check_pair()
