#!/usr/bin/env python3
"""GPU box: PCIe-inclusive rate of the streaming ABI (ro_stft_push -> kernels -> ro_stft_fetch of the
recorder band), N=32768 / 75 % overlap.  Reported in DESIGN.md; never the bench headline."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ro = importlib.import_module("radio-observer_amd")
bins, overlap, hop = 32768, 24576, 8192
rows = 4096
rng = np.random.default_rng(0)
iq = rng.standard_normal((bins + (rows - 1) * hop, 2)).astype(np.float32)
for cols, name in ((bins, "full rows (128 KiB/row back to the host)"), (2048, "recorder band 9-12 kHz (8 KiB/row)")):
    with ro.Stft(bins=bins, overlap=overlap, max_batch_rows=256, tile=None if cols == bins else (22528, cols)) as st:
        t0 = time.perf_counter()
        got = 0
        for i in range(0, iq.shape[0], 1 << 20):
            st.push(iq[i:i + (1 << 20)])
            while True:
                _, r, _ = st.fetch(256, first_col=22528 if cols != bins else 0, cols=cols)
                if len(r) == 0:
                    break
                got += len(r)
        st.flush()
        while True:
            _, r, _ = st.fetch(256, first_col=22528 if cols != bins else 0, cols=cols)
            if len(r) == 0:
                break
            got += len(r)
        dt = time.perf_counter() - t0
    print("%s: %d rows in %.3f s = %.3g rows/s (%.2f GB/s of samples in)" % (name, got, dt, got / dt, iq.nbytes / dt / 1e9))
