#!/usr/bin/env python3
"""Index algebra of csrc/ro_fourstep.hip (the four-step large transforms) emulated on the CPU, like emu32k.py for the
row kernel: the column kernel's sample / window / exchange / scratch addresses, the scratch order both kernels
agree on, the row kernel's loads, its three twiddle tables, and the columns its image read-back stores to -- all
against numpy's FFT of a windowed row.  Passes 1 and 2 themselves (planar butterflies, exchange 2, the image) are the
row kernel's and are checked by emu32k.py; here they are numpy FFTs over the slots a thread holds."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from emu32k import bitrev, RQ  # noqa: E402

T = 1024


def zfloat(k1, n2, comp):
    """float index of component comp (0 re, 1 im) of Z[k1][n2] inside a stream row's scratch: the planar pairs"""
    a, b = n2 & 31, n2 >> 5
    i, p = b >> 1, b & 1
    return k1 * 2048 + (i * 32 + a) * 4 + 2 * comp + p


def tables(bins, window):
    """fourstep_tables of the .hip, restated"""
    n1 = bins // 1024
    r2, c = n1 // 32, 1024 // (n1 // 32)
    wa = np.zeros(bins)
    for cg in range(r2):
        for q in range(8):
            t = np.arange(1024)
            m, col = t // c, t % c
            for e in range(4):
                l = 4 * q + e
                wa[((cg * 8 + q) * 1024 + t) * 4 + e] = window[1024 * (m + r2 * l) + c * cg + col]
    tw_a = np.exp(-2j * np.pi * np.outer(np.arange(32), np.arange(r2)) / n1)             # [k_l][m]
    j = 1 << np.arange(5)
    tw_b1 = np.exp(-2j * np.pi * 32 * np.outer(np.arange(n1), j) / bins)                  # [k1][j]
    tw_b2 = np.exp(-2j * np.pi * np.outer(np.arange(n1), j) / bins)
    tw_r = np.exp(-2j * np.pi * np.outer(np.arange(32), j) / 1024)
    return wa, tw_a, tw_b1, tw_b2, tw_r


def check(bins, seed):
    n1 = bins // 1024
    R2 = n1 // 32
    C, SETS = T // R2, 32 // R2
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(bins) + 1j * rng.standard_normal(bins)
    win = rng.standard_normal(bins)
    want = np.fft.fftshift(np.fft.fft(x * win))
    Ztrue = np.fft.fft((x * win).reshape(n1, 1024), axis=0)                              # [k1][n2]
    wa, tw_a, tw_b1, tw_b2, tw_r = tables(bins, win)
    zmax = np.abs(Ztrue).max()

    # ---- the column kernel, a few blocks (column groups cg of the row)
    for cg in (1, R2 - 1):
        seen = set()
        v = np.zeros((T, 32), dtype=complex)
        for tid in range(T):
            grp, col = tid // C, tid % C
            vo = 1024 * grp + C * cg + col                                               # samples
            l = np.arange(32)
            s_idx = vo + l * 1024 * R2
            c = wa[((cg * 8 + (l >> 2)) * T + tid) * 4 + (l & 3)]
            assert (c == win[s_idx]).all()
            v[tid] = x[s_idx] * c
            seen.update(s_idx.tolist())
        assert len(seen) == 32768
        y = np.fft.fft(v, axis=1)                                                        # [tid][k_l]
        plane = np.zeros(32 * T, dtype=complex)
        for k in range(32):
            plane[k * T + np.arange(T)] = y[:, k]
        scratch = {}
        for tid in range(T):
            grp, col, wave, lane = tid // C, tid % C, tid >> 6, tid & 63
            i_quad = (C // 64) * cg + ((wave & 1) if C == 128 else 0)
            for h in range(SETS):
                kl = SETS * grp + h
                u = np.array([plane[kl * T + m * C + col] for m in range(R2)]) * tw_a[kl]
                z = np.fft.fft(u)                                                        # [k_m]
                for km in range(R2):
                    k1 = kl + 32 * km
                    n2 = C * cg + col
                    assert abs(z[km] - Ztrue[k1, n2]) < 1e-9 * zmax
                    # what the lane stores after the permlane32_swap of (im, re): two floats at zo
                    zo = (i_quad * 32 + (lane & 31)) * 4 + (2 if lane < 32 else 0)
                    mate = n2 + 32 if lane < 32 else n2 - 32                             # lane c +- 32 of the same wave
                    if lane < 32:
                        pair = (Ztrue[k1, n2].imag, Ztrue[k1, mate].imag)
                        where = (zfloat(k1, n2, 1), zfloat(k1, mate, 1))
                    else:
                        pair = (Ztrue[k1, mate].real, Ztrue[k1, n2].real)
                        where = (zfloat(k1, mate, 0), zfloat(k1, n2, 0))
                    assert where == (k1 * 2048 + zo, k1 * 2048 + zo + 1), (tid, h, km, where, zo)
                    for wq, val in zip(where, pair):
                        assert wq not in scratch
                        scratch[wq] = val
        assert len(scratch) == 2 * 32768
    print("bins = %d: column kernel: samples, window table, exchange and scratch addresses" % bins)

    # ---- the row kernel, a few blocks g (rows k1 = 32 g ..)
    zs = np.zeros(n1 * 2048)
    k1s, n2s = np.meshgrid(np.arange(n1), np.arange(1024), indexing="ij")
    zs[zfloat(k1s, n2s, 0)] = Ztrue.real
    zs[zfloat(k1s, n2s, 1)] = Ztrue.imag
    G = n1 // 32
    out = np.full(bins, np.nan, dtype=complex)
    for g in (0, G - 1):
        blk = zs[g * 32 * 2048:(g + 1) * 32 * 2048]
        y1 = np.zeros((T, 32), dtype=complex)
        for tau in range(T):
            wave, lane = tau >> 6, tau & 63
            zo = (2 * wave + (lane >> 5)) * 2048 + (2 * (lane & 15) + ((lane >> 4) & 1)) * 4
            k1 = 32 * g + 2 * wave + (lane >> 5)
            a = 2 * (lane & 15) + ((lane >> 4) & 1)
            slots = np.zeros(32, dtype=complex)
            for i in range(16):
                q = blk[zo + i * 128:zo + i * 128 + 4]                                    # 512 bytes per i
                slots[2 * i], slots[2 * i + 1] = q[0] + 1j * q[2], q[1] + 1j * q[3]
            assert np.allclose(slots, Ztrue[k1, a + 32 * np.arange(32)])
            w1 = tw_b1[k1]                                                               # powers 2^j of the stage twiddle
            assert np.allclose(w1, w1[0] ** (1 << np.arange(5)))
            y1[tau] = np.fft.fft(slots * w1[0] ** np.arange(32))                         # slot k_r
        # exchange 2 (emu32k.py): pass-2 thread (wave, lane') holds, for its row rho = 2 wave + (lane' & 1) and
        # k_r = ((lane' >> 1) + 4 (wave >> 1)) & 31, the values of the 32 a
        img = np.zeros(32 * RQ, dtype=complex)
        for tau in range(T):
            wave, lane = tau >> 6, tau & 63
            kbp, kr = lane & 1, ((lane >> 1) + 4 * (wave >> 1)) & 31
            rho = 2 * wave + kbp
            k1 = 32 * g + rho
            src = np.array([y1[64 * wave + (a >> 1) + 16 * (a & 1) + 32 * kbp, kr] for a in range(32)])
            w2 = tw_r[kr] * tw_b2[k1]
            y2 = np.fft.fft(src * w2[0] ** np.arange(32))                                # slot k_c
            for kc in range(32):
                img[RQ * kc + 64 * wave + lane] = y2[kc]
        for q in range(8):
            for tid in range(T):
                base = RQ * (tid >> 8) + 128 * (tid & 7) + 2 * ((((tid & 255) >> 3) - 4 * (tid & 7)) & 31) + 4 * RQ * q
                vo = 4 * (tid & 7) + n1 * ((tid & 255) >> 3) + n1 * 32 * (tid >> 8)
                so = n1 * 128 * ((q + 4) & 7)
                col0 = 32 * g + vo + so
                assert col0 + 3 < bins and (vo + so + 3) < bins - 32 * g
                out[col0:col0 + 4] = [img[base], img[base + 1], img[base + 64], img[base + 65]]
    done = ~np.isnan(out)
    assert done.sum() == 2 * 32768
    assert np.abs(out[done] - want[done]).max() < 1e-9 * np.abs(want).max()
    # eight lanes = one 128-byte line of the row
    for tid in range(0, T, 8):
        vo = 4 * (tid & 7) + n1 * ((tid & 255) >> 3) + n1 * 32 * (tid >> 8)
        assert vo % 32 == 0
    print("bins = %d: row kernel: scratch loads, twiddle tables, fft-shifted columns of the read-back" % bins)


if __name__ == "__main__":
    check(262144, 3)
    check(524288, 4)
