#!/usr/bin/env python3
"""Index algebra of stft32k_kernel (N = 32768, 1024 threads) emulated on the CPU.

Every LDS address map, lane map and twiddle index of the kernel is restated here with numpy and checked
(a) end to end against numpy's FFT and (b) for LDS bank conflicts per wave-instruction, before any of it
is run on a GPU.  Constants must match csrc/ro_stft32k.hip (RP, ROWP, PLANE_ODD)."""
import numpy as np

N = 32768
RQ = 1025          # floats per image row q (16 waves x 64 lanes + 1: odd, so a column read walks the banks)
def cell(q, w, lane):       # the one LDS layout of the kernel: row q, territory of wave w, lane
    return RQ * q + 64 * w + lane
def rot(w):                 # per-wave rotation of the pass-2 lane map (keeps the row read-back conflict-free)
    return 4 * (w >> 1)

def column(t):     # stage-0 column of thread position t (paired 16-byte loads + permlane32 swap)
    w, l = t >> 6, t & 63
    return 64 * w + 2 * (l & 31) + (l >> 5)

def lane_p1(kb, a):   # lane of pass-1 thread (kb, a) inside its wave
    return (a >> 1) + 16 * kb + 32 * (a & 1)

def x1_cell(k0, t):   # exchange 1: slot k0 of stage-0 thread t -> row (writer wave + 16 (k0 & 1)), territory k0 >> 1
    return cell((t >> 6) + 16 * (k0 & 1), k0 >> 1, t & 63)

def banks_ok(addrs_by_lane, group=32, nb=32):
    for g in range(0, 64, group):
        seen = {}
        for a in addrs_by_lane[g:g + group]:
            b = a % nb
            if b in seen and seen[b] != a:
                return False
            seen[b] = a
    return True

def main():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    want = np.fft.fft(x)
    lds = np.zeros(32 * RQ, dtype=complex)
    T = 1024
    y0 = np.zeros((T, 32), dtype=complex)
    for t in range(T):
        y0[t] = np.fft.fft(x[column(t) + 1024 * np.arange(32)])
    used = set()
    for k0 in range(32):
        for t in range(T):
            c = x1_cell(k0, t)
            assert c not in used
            used.add(c)
            lds[c] = y0[t, k0]
    ok = True
    v1 = np.zeros((T, 32), dtype=complex)
    for w in range(16):
        for b in range(32):
            addrs = []
            for lam in range(64):
                kb = (lam >> 4) & 1
                base = cell(16 * kb, w, (lam & 15) + 32 * (lam >> 5))
                ad = base + RQ * (b >> 1) + 16 * (b & 1)
                addrs.append(ad)
                a = 2 * (lam & 15) + (lam >> 5)
                assert lane_p1(kb, a) == lam
                v1[64 * w + lam, b] = lds[ad]
                # territory: only cells of wave w
                assert (ad % RQ) // 64 == w
            ok &= banks_ok(addrs)
    print("exchange 1 gathers conflict-free:", ok)
    y1 = np.zeros((T, 32), dtype=complex)
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        k0 = 2 * w + ((lam >> 4) & 1)
        y1[tau] = np.fft.fft(v1[tau] * np.exp(-2j * np.pi * k0 * np.arange(32) / 1024))
    lds[:] = 0
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        for k1 in range(32):
            lds[cell(k1, w, lam)] = y1[tau, k1]
    v2 = np.zeros((T, 32), dtype=complex)
    ok = True
    for w in range(16):
        for a in range(32):
            addrs = []
            for lam in range(64):
                k1, kb = ((lam & 31) + rot(w)) & 31, lam >> 5
                ad = cell(k1, w, lane_p1(kb, a))
                addrs.append(ad)
                v2[64 * w + lam, a] = lds[ad]
            ok &= banks_ok(addrs)
    print("exchange 2 gathers conflict-free:", ok)
    y2 = np.zeros((T, 32), dtype=complex)
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        k1, kb = ((lam & 31) + rot(w)) & 31, lam >> 5
        kp = 2 * w + kb + 32 * k1
        y2[tau] = np.fft.fft(v2[tau] * np.exp(-2j * np.pi * kp * np.arange(32) / N))
        assert np.abs(y2[tau] - want[kp + 1024 * np.arange(32)]).max() < 1e-6 * np.abs(want).max()
    print("bins match numpy fft")
    lds[:] = 0
    for tau in range(T):
        w, lam = tau >> 6, tau & 63
        for k2 in range(32):
            lds[cell(k2, w, lam)] = y2[tau, k2]
    out = np.zeros(N, dtype=complex)
    ok = True
    def rb(tid, q, i):
        mg = tid + T * q
        r, m = mg >> 8, mg & 255
        w = 2 * (m & 7) + (i >> 1)
        return cell(r, w, (((m >> 3) - rot(w)) & 31) + 32 * (i & 1)), (1024 * r + 4 * m + i + N // 2) % N
    for q in range(8):
        for tid in range(T):
            for i in range(4):
                ad, col = rb(tid, q, i)
                out[col] = lds[ad]
        for wv in range(16):
            for i in range(4):
                ok &= banks_ok([rb(64 * wv + lam, q, i)[0] for lam in range(64)])
    assert np.allclose(out, np.fft.fftshift(want))
    print("read-back = fft-shifted row; conflict-free:", ok)
    # base + immediate form of the read-back address: base(tid) + imm(q, i)
    for tid in range(T):
        b0 = rb(tid, 0, 0)[0]
        for q in range(8):
            for i in range(4):
                assert rb(tid, q, i)[0] - b0 == RQ * 4 * q + 64 * (i >> 1) + 32 * (i & 1)
    print("read-back address = base(tid) + 4100 q + 64 (i>>1) + 32 (i&1)")
    # add-TID reach and the M0 / offset split (16 bits each)
    HB = 61568
    for w in range(16):
        for k0 in range(32):
            m0 = 4 * RQ * w + (HB - 57536 if False else 0)
        for q in range(32):
            m0 = 256 * w + (HB if q >= 16 else 0)
            off = 4 * RQ * q - (HB if q >= 16 else 0)
            assert 0 <= m0 <= 65535 and 0 <= off <= 65535 and m0 + off == 4 * cell(q, w, 0)
        for k0 in range(32):
            hb = 4032 if k0 & 1 else 0
            m0 = 4 * RQ * w + hb
            off = 4 * (RQ * 16 * (k0 & 1) + 64 * (k0 >> 1)) - hb
            assert 0 <= m0 <= 65535 and 0 <= off <= 65535 and m0 + off == 4 * x1_cell(k0, 64 * w), (w, k0, m0, off)
    print("M0 / offset splits fit 16 bits; LDS bytes:", 32 * RQ * 4)

if __name__ == "__main__":
    main()
