#!/usr/bin/env python3
"""Index algebra of the wave-local STFT kernel family (N = 1024 S: S = 32 -> 32768, 16 -> 16384, 8 -> 8192) on the CPU:
every LDS address map, lane map and twiddle index restated with numpy, checked end to end against numpy's FFT and for
LDS bank conflicts per wave-instruction.  Constants must match csrc/ro_stft32k.hip."""
import sys
import numpy as np


def banks_ok(addrs, group=32, nb=32):
    for g in range(0, 64, group):
        seen = {}
        for a in addrs[g:g + group]:
            b = a % nb
            if b in seen and seen[b] != a:
                return False
            seen[b] = a
    return True


def run(S):
    N, TL = 1024 * S, 32 * S
    W, G, H = TL // 64, 64 // S, S // 2          # waves, k0 values per wave, lanes of a group in one half-wave
    NB2 = 32 // S                                # radix-S butterflies per thread in pass 2
    RQ = 64 * W + 1

    def cell(q, w, lane):
        return RQ * q + 64 * w + lane

    def column(t):
        w, l = t >> 6, t & 63
        return 64 * w + 2 * (l & 31) + (l >> 5)

    def lane_p1(j, a):                           # pass-1 lane of (k0 = G w + j, a)
        return (a >> 1) + H * j + 32 * (a & 1)

    def rot(w):                                  # per-wave rotation of the pass-2 lane map (read-back conflicts at S = 32)
        return 4 * (w >> 1) if S == 32 else 0

    def pi(k1):                                  # row of slot k1 in exchange 2 (column reads of a group walk the banks)
        i, m = k1 % S, k1 // S
        return (i % H) + 16 * (i // H) + H * m

    def p2(lam):                                 # pass-2 lane -> (j', i): k0 = G w + j', k1 = i + S m
        return lam // S, lam % S

    rng = np.random.default_rng(S)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    want = np.fft.fft(x)
    lds = np.zeros(32 * RQ, dtype=complex)
    # ---- pass 0
    y0 = np.stack([np.fft.fft(x[column(t) + TL * np.arange(32)]) for t in range(TL)])
    used = set()
    for k0 in range(32):
        for t in range(TL):
            c = cell((t >> 6) + W * (k0 % G), k0 // G, t & 63)
            assert c not in used
            used.add(c)
            lds[c] = y0[t, k0]
    # ---- exchange 1 reads
    v1 = np.zeros((TL, 32), dtype=complex)
    ok = True
    for w in range(W):
        for b in range(32):
            addrs = []
            for lam in range(64):
                j, ah, par = (lam & 31) // H, (lam & 31) % H, lam >> 5
                a = 2 * ah + par
                assert lane_p1(j, a) == lam
                ad = cell((b // G) + W * j, w, ah + H * (b % G) + 32 * par)
                # the column this cell holds
                wr, l = (ad % RQ) // 64, None
                addrs.append(ad)
                v1[64 * w + lam, b] = lds[ad]
            ok &= banks_ok(addrs)
    # check values: thread (k0, a) slot b must be y0[column a + S b][k0]
    pos = {column(t): t for t in range(TL)}
    for w in range(W):
        for lam in range(64):
            j, ah, par = (lam & 31) // H, (lam & 31) % H, lam >> 5
            a, k0 = 2 * ah + par, G * w + j
            for b in range(32):
                assert v1[64 * w + lam, b] == y0[pos[a + S * b], k0], (w, lam, b)
    print("S=%d exchange 1 gathers conflict-free: %s" % (S, ok))
    # ---- pass 1
    y1 = np.zeros((TL, 32), dtype=complex)
    for tau in range(TL):
        w, lam = tau >> 6, tau & 63
        k0 = G * w + (lam & 31) // H
        y1[tau] = np.fft.fft(v1[tau] * np.exp(-2j * np.pi * k0 * np.arange(32) / 1024))
    lds[:] = 0
    for tau in range(TL):
        for k1 in range(32):
            lds[cell(pi(k1), tau >> 6, tau & 63)] = y1[tau, k1]
    assert sorted(pi(k) for k in range(32)) == list(range(32))
    # ---- exchange 2 + pass 2: lane (j', i): k0 = G w + j', k1 = ((i + rot) mod S) + S m, m < NB2
    ok = True
    out_bins = {}
    v2 = np.zeros((TL, NB2, S), dtype=complex)
    for w in range(W):
        for m in range(NB2):
            for a in range(S):
                addrs = []
                for lam in range(64):
                    j, il = p2(lam)
                    i = (il + rot(w)) % S
                    k1 = i + S * m
                    ad = cell(pi(k1), w, lane_p1(j, a))
                    addrs.append(ad)
                    v2[64 * w + lam, m, a] = lds[ad]
                ok &= banks_ok(addrs)
    print("S=%d exchange 2 gathers conflict-free: %s" % (S, ok))
    y2 = np.zeros((TL, NB2, S), dtype=complex)
    for tau in range(TL):
        w, lam = tau >> 6, tau & 63
        j, il = p2(lam)
        i = (il + rot(w)) % S
        k0 = G * w + j
        for m in range(NB2):
            kp = k0 + 32 * (i + S * m)
            y2[tau, m] = np.fft.fft(v2[tau, m] * np.exp(-2j * np.pi * kp * np.arange(S) / N))
            assert np.abs(y2[tau, m] - want[kp + 1024 * np.arange(S)]).max() < 1e-6 * np.abs(want).max()
    print("S=%d bins match numpy fft" % S)
    # ---- image: slot (m, k2) -> row m S + k2; read-back of 4 consecutive bins per lane
    lds[:] = 0
    where = {}
    for tau in range(TL):
        w, lam = tau >> 6, tau & 63
        j, il = p2(lam)
        i = (il + rot(w)) % S
        for m in range(NB2):
            for k2 in range(S):
                c = cell(m * S + k2, w, lam)
                lds[c] = y2[tau, m, k2]
                where[G * w + j + 32 * (i + S * m) + 1024 * k2] = c
    # read-back: chunk q, thread tid: mg = tid + TL q -> bins 4 mg .. 4 mg + 3
    ok, worst = True, 1
    out = np.zeros(N, dtype=complex)
    chunks = N // 4 // TL
    for q in range(chunks):
        for wv in range(W):
            for i4 in range(4):
                addrs = []
                for lam in range(64):
                    mg = 64 * wv + lam + TL * q
                    ad = where[4 * mg + i4]
                    addrs.append(ad)
                    out[(4 * mg + i4 + N // 2) % N] = lds[ad]
                ok &= banks_ok(addrs)
                for g in (0, 32):
                    worst = max(worst, max(np.bincount([a % 32 for a in set(addrs[g:g + 32])])))
    assert np.allclose(out, np.fft.fftshift(want))
    print("S=%d read-back = fft-shifted row; conflict-free: %s (worst %d-way)" % (S, ok, worst))
    # closed form of the read-back address
    bad = 0
    for q in range(chunks):
        for tid in range(TL):
            mg = tid + TL * q
            for i4 in range(4):
                k = 4 * mg + i4
                k2, beta = k >> 10, k & 1023
                k0, k1 = beta & 31, beta >> 5
                w, j = k0 // G, k0 % G
                m, i = k1 // S, k1 % S
                il = (i - rot(w)) % S
                lam = il + S * j
                bad += cell(m * S + k2, w, lam) != where[k]
    print("S=%d closed-form image address: %s; LDS bytes %d" % (S, bad == 0, 32 * RQ * 4))
    return ok


if __name__ == "__main__":
    for S in ([int(x) for x in sys.argv[1:]] or (32, 16, 8)):
        run(S)
