"""The index algebra of stft32k_kernel's one LDS layout, replayed on the CPU: tools/r4/emu32k.py walks
the kernel's three passes with the kernel's own lane / slot / cell maps (and its window-table order) on a random row
and compares with numpy's fft; on the way it asserts that every ds_read_b64 is aligned and bank-conflict-free and that
every add-TID write splits into a 16-bit M0 and a 16-bit immediate.  It also runs the planar butterflies of
csrc/ro_fft_planar.h -- with the VOP3P modifier strings parsed out of that header -- against the 32-point DFT.  The kernel itself is checked on the GPU (tests/test_gpu_*.py); this keeps the derivation it
was written from under test, so a change of the layout is made there first."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args, where="r3"):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", where, script), *args], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_stft32k_layout_reproduces_the_fft_without_bank_conflicts():
    out = run("emu32k.py", where="r4")
    assert "planar radix-32 stage (modifier strings from ro_fft_planar.h) == DFT32" in out
    assert "exchange 1 reads (ds_read_b64) conflict-free: True" in out
    assert "exchange 2 reads (ds_read_b64) conflict-free: True" in out
    assert "bins match numpy fft" in out
    assert "= fft-shifted row; conflict-free: True" in out
    assert "ImageRow(c) = column c" in out
    assert "LDS bytes: 131328" in out


def test_fourstep_maps_reproduce_the_fft():
    """tools/r4/emu_four.py: the column kernel's sample / window / exchange / scratch addresses, the scratch order, the
    row kernel's loads, twiddle tables and output columns of csrc/ro_fourstep.hip at 262144, 524288 and 1048576 bins"""
    out = run("emu_four.py", where="r4")
    for bins in (262144, 524288, 1048576):
        assert "bins = %d: column kernel: samples, window table, exchange and scratch addresses" % bins in out
        assert "bins = %d: row kernel: scratch loads, twiddle tables, fft-shifted columns of the read-back; LDS reads conflict-free" % bins in out


def test_the_general_form_agrees_at_32_points_per_wave_column():
    """tools/r3/emu_wl.py is the derivation of round 3's layout (ds_read_b32 gathers, rows of 1025 floats) for 32.32.S;
    it stays as the record of the experiment in tools/r3/ro_stft_wl.hip"""
    out = run("emu_wl.py", "32")
    assert "S=32 read-back = fft-shifted row; conflict-free: True" in out
    assert "S=32 closed-form image address: True; LDS bytes 131200" in out


def test_f64r_maps_reproduce_the_fft_without_bank_conflicts():
    """tools/r6/emu_f64r.py: the register-resident FP64 kernel (csrc/ro_f64reg.hip) -- twisted radix-16 butterflies with merged
    twiddles, every thread map, exchange cell, lane swap, table index and read-out column for each (M, D) the library
    routes to it -- against numpy's FFT, with every LDS access pattern conflict-free; and the constants it was run with
    are the kernel's"""
    out = run("emu_f64r.py", where="r6")
    assert "butterflies ok" in out
    for m, d in ((4096, 1), (8192, 1), (16384, 1), (16384, 2), (16384, 4)):
        assert "M = %5d  D = %d" % (m, d) in out
    for bins in (256, 512, 1024, 2048):                  # rows batched into the 4096-point workgroup
        assert "bins %4d, %2d rows per workgroup" % (bins, 4096 // bins) in out
    assert "all maps ok, every LDS access conflict-free" in out
    src = open(os.path.join(ROOT, "radio-observer_amd", "csrc", "ro_f64reg.hip")).read()
    assert "ST = T + 16 * R3" in src and "S2 = Q + (R3 == 1 ? 1 : 2)" in src
