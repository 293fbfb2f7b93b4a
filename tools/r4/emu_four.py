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
from emu32k import bitrev, banks_ok_b64  # noqa: E402

WAVES = 16                                       # csrc/ro_fourstep.hip: RO_FOUR_WAVES (8: the two-workgroups-per-CU experiment)
T, RQ, BROWS, ROT = 64 * WAVES, 64 * WAVES + 2, 2 * WAVES, 64 // WAVES


def set_waves(w):
    global WAVES, T, RQ, BROWS, ROT
    WAVES = w
    T, RQ, BROWS, ROT = 64 * w, 64 * w + 2, 2 * w, 64 // w


def column(C, cg, lam):
    """fourstep_column of the .hip"""
    if C >= 64:
        return C * cg + lam
    return (16 * (cg & 1) + (lam & 15)) + 32 * (2 * (cg >> 1) + (lam >> 4))


def zfloat(k1, n2, comp):
    """float index of component comp (0 re, 1 im) of Z[k1][n2] inside a stream row's scratch: the planar pairs"""
    a, b = n2 & 31, n2 >> 5
    i, p = b >> 1, b & 1
    return k1 * 2048 + (i * 32 + a) * 4 + 2 * comp + p


def tables(bins, window):
    """fourstep_tables of the .hip, restated"""
    n1 = bins // 1024
    r2 = n1 // 32
    c = T // r2
    wa = np.zeros(bins)
    for cg in range(1024 // c):
        for q in range(8):
            t = np.arange(T)
            m, lam = t // c, t % c
            col = np.array([column(c, cg, int(x)) for x in lam])
            for e in range(4):
                l = 4 * q + e
                wa[((cg * 8 + q) * T + t) * 4 + e] = window[1024 * (m + r2 * l) + col]
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
    GROUPS = 1024 // C
    for cg in (1, 2, GROUPS - 1):
        seen = set()
        v = np.zeros((T, 32), dtype=complex)
        for tid in range(T):
            grp, lam = tid // C, tid % C
            vo = 1024 * grp + column(C, cg, lam)                                         # samples
            l = np.arange(32)
            s_idx = vo + l * 1024 * R2
            c = wa[((cg * 8 + (l >> 2)) * T + tid) * 4 + (l & 3)]
            assert (c == win[s_idx]).all()
            v[tid] = x[s_idx] * c
            seen.update(s_idx.tolist())
        assert len(seen) == 32 * T
        y = np.fft.fft(v, axis=1)                                                        # [tid][k_l]
        plane = np.zeros(32 * T, dtype=complex)
        for k in range(32):
            plane[k * T + np.arange(T)] = y[:, k]
        scratch = {}
        for tid in range(T):
            grp, lam, wave, lane = tid // C, tid % C, tid >> 6, tid & 63
            n2 = column(C, cg, lam)
            p0 = ((n2 >> 5) & 1) == 0
            # the mate sits 32 (C >= 64: v_permlane32_swap) or 16 (C = 32: v_permlane16_swap, odd <-> even rows of 16) lanes away
            mate_lane = lane + (32 if C >= 64 else 16) * (1 if p0 else -1)
            assert 0 <= mate_lane < 64 and (p0 == (lane < 32) if C >= 64 else p0 == (((lane >> 4) & 1) == 0))
            mate_tid = 64 * wave + mate_lane
            assert mate_tid // C == grp and column(C, cg, mate_tid % C) == n2 + (32 if p0 else -32)
            for h in range(SETS):
                kl = SETS * grp + h
                u = np.array([plane[kl * T + m * C + lam] for m in range(R2)]) * tw_a[kl]
                z = np.fft.fft(u)                                                        # [k_m]
                for km in range(R2):
                    assert abs(z[km] - Ztrue[kl + 32 * km, n2]) < 1e-9 * zmax
                # Two rows k_m = 2 kp (A), 2 kp + 1 (B) leave together: behind the lane swaps (vdst = B, src = A, one per
                # component) a lane of p = 0 holds (B, B') and stores the whole quad of row B, a lane of p = 1 that of
                # row A: 16 bytes at zo + soffset
                for kp in range(R2 // 2):
                    k1 = kl + 32 * (2 * kp + (1 if p0 else 0))
                    lo, hi = (n2, n2 + 32) if p0 else (n2 - 32, n2)                      # the columns of p = 0, p = 1
                    quad = (Ztrue[k1, lo].real, Ztrue[k1, hi].real, Ztrue[k1, lo].imag, Ztrue[k1, hi].imag)
                    where = (zfloat(k1, lo, 0), zfloat(k1, hi, 0), zfloat(k1, lo, 1), zfloat(k1, hi, 1))
                    zo = ((n2 >> 6) * 32 + (n2 & 31)) * 4 + (32 * 2048 if p0 else 0)      # floats; + the half-wave's rows if C = 32
                    so = (kl + 64 * kp) * 2048
                    assert where == tuple(zo + so + j for j in range(4)), (tid, h, kp, where, zo, so)
                    for wq, val in zip(where, quad):
                        assert wq not in scratch
                        scratch[wq] = val
        assert len(scratch) == 2 * 32 * T
    print("bins = %d: column kernel: samples, window table, exchange and scratch addresses" % bins)

    # ---- the row kernel, a few blocks g (rows k1 = 32 g ..)
    zs = np.zeros(n1 * 2048)
    k1s, n2s = np.meshgrid(np.arange(n1), np.arange(1024), indexing="ij")
    zs[zfloat(k1s, n2s, 0)] = Ztrue.real
    zs[zfloat(k1s, n2s, 1)] = Ztrue.imag
    G = n1 // BROWS

    def rb(tid):                                     # the read-back's LDS address and store offset (floats) of thread tid
        t_hi, rj, r_kr = tid // (T // 4), tid % (WAVES // 2), (tid % (T // 4)) // (WAVES // 2)
        return RQ * t_hi + 128 * rj + 2 * ((r_kr - ROT * rj) & 31)

    def out_vo(tid):
        t_hi, rj, r_kr = tid // (T // 4), tid % (WAVES // 2), (tid % (T // 4)) // (WAVES // 2)
        return 4 * rj + n1 * r_kr + n1 * 32 * t_hi

    out = np.full(bins, np.nan, dtype=complex)
    ok2 = okrb = True
    for g in (0, 5, G - 1):
        blk = zs[g * BROWS * 2048:(g + 1) * BROWS * 2048]
        y1 = np.zeros((T, 32), dtype=complex)
        for tau in range(T):
            wave, lane = tau >> 6, tau & 63
            zo = (2 * wave + (lane >> 5)) * 2048 + (2 * (lane & 15) + ((lane >> 4) & 1)) * 4
            k1 = BROWS * g + 2 * wave + (lane >> 5)
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
            kbp, kr = lane & 1, ((lane >> 1) + ROT * (wave >> 1)) & 31
            rho = 2 * wave + kbp
            k1 = BROWS * g + rho
            src = np.array([y1[64 * wave + (a >> 1) + 16 * (a & 1) + 32 * kbp, kr] for a in range(32)])
            w2 = tw_r[kr] * tw_b2[k1]
            y2 = np.fft.fft(src * w2[0] ** np.arange(32))                                # slot k_c
            for kc in range(32):
                img[RQ * kc + 64 * wave + lane] = y2[kc]
        # exchange 2's ds_read_b64 (rows k_r of the wave's territory) are conflict-free with the rotation by ROT (w >> 1)
        for wave in range(T // 64):
            for i in range(16):
                ok2 &= banks_ok_b64([RQ * (((lane >> 1) + ROT * (wave >> 1)) & 31) + 64 * wave + 32 * (lane & 1) + 2 * ((i >> 1) + 8 * (i & 1))
                                     for lane in range(64)])
        for q in range(8):
            for tid in range(T):
                base = rb(tid) + 4 * RQ * q
                vo = out_vo(tid)
                so = n1 * 128 * ((q + 4) & 7)
                col0 = BROWS * g + vo + so
                assert col0 + 3 < bins and (vo + so + 3) < bins - BROWS * g
                out[col0:col0 + 4] = [img[base], img[base + 1], img[base + 64], img[base + 65]]
            for wave in range(T // 64):
                for half in (0, 64):
                    okrb &= banks_ok_b64([rb(t) + 4 * RQ * q + half for t in range(64 * wave, 64 * wave + 64)])
    assert ok2 and okrb, (ok2, okrb)
    HBQ = 61692 if 4 * RQ * 31 > 65535 else 0        # ro_k32_lds.h's HB
    assert 4 * RQ * 15 <= 65535 and 0 <= 4 * RQ * 16 - HBQ and 4 * RQ * 31 - HBQ <= 65535 and 15 * 256 + HBQ <= 65535
    done = ~np.isnan(out)
    assert done.sum() == 3 * 32 * T
    assert np.abs(out[done] - want[done]).max() < 1e-9 * np.abs(want).max()
    # WAVES / 2 lanes = BROWS neighbouring columns of the row (sixteen waves: one 128-byte line)
    for tid in range(0, T, WAVES // 2):
        assert out_vo(tid) % BROWS == 0
    print("bins = %d: row kernel: scratch loads, twiddle tables, fft-shifted columns of the read-back; LDS reads conflict-free" % bins)


if __name__ == "__main__":
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "radio-observer_amd", "csrc", "ro_fourstep.hip")).read()
    assert "#define RO_FOUR_WAVES %d" % WAVES in src and "constexpr int WAVES = RO_FOUR_WAVES" in src and "constexpr int RQ = 64 * WAVES + 2;" in src and "constexpr int ROT = 64 / WAVES;" in src
    check(262144, 3)
    check(524288, 4)
    check(1048576, 5)
    set_waves(8)
    check(524288, 6)
    print("(the last one with RO_FOUR_WAVES = 8)")
