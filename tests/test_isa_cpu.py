"""Static checks on the gfx950 ISA hipcc produces for the kernels (cross-compiled here, no GPU needed).

  * no kernel uses scratch (a spill in the 1024-thread STFT kernel costs far more than it looks)
  * between two consecutive s_barrier of a kernel there is LDS traffic: hipcc once sank an exchange's LDS gather
    loads below the exchange's closing barrier (two barriers back to back, the reads after them) and rows went wrong
    at random under load; ro_kernels.hip's wg_sync() exists to prevent that
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "radio-observer_amd", "csrc", "ro_kernels.hip")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    out = tmp_path_factory.mktemp("isa") / "ro_kernels.s"
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only",
           SRC, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def kernels(text):
    """kernel name -> its instruction lines (from the label to s_endpgm)"""
    out, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^(_ZN2ro\w+):", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if "s_endpgm" in line:
                out[name] = body
                name = None
    return out


def test_kernels_have_no_scratch(isa):
    # one metadata entry per kernel; .name precedes .private_segment_fixed_size inside it
    entries = re.findall(r"\.name:\s*(\S+)[\s\S]*?\.private_segment_fixed_size:\s*(\d+)", isa)
    assert len(entries) >= 20
    bad = [(n, int(s)) for n, s in entries if int(s) != 0]
    assert not bad, bad


def test_lds_traffic_between_consecutive_barriers(isa):
    ks = kernels(isa)
    stft = [k for k in ks if "stft_kernel" in k]
    # 8 plans x 2 sample formats x {magnitudes, spectra} + the one-kernel large transform on the N = 32768 plan x 2 formats
    assert len(stft) == 34
    for name, body in ks.items():
        seen, lds = False, False
        for i, line in enumerate(body):
            if "s_barrier" in line:
                assert not seen or lds, "adjacent barriers in %s at +%d" % (name, i)
                seen, lds = True, False
            elif re.search(r"\bds_(read|write|add)|buffer_load.*\blds\b", line):
                lds = True


def test_addtid_writes_are_waited_for_before_the_barrier(isa):
    """ds_write_addtid_b32 sits in inline asm, which hipcc's waitcnt insertion does not see: every s_barrier that
    follows such writes needs an explicit s_waitcnt lgkmcnt(0) in between, or other waves read the image too early."""
    ks = kernels(isa)
    checked = 0
    for name, body in ks.items():
        pending = False
        for i, line in enumerate(body):
            if "ds_write_addtid_b32" in line:
                pending = True
                checked += 1
            elif pending and re.search(r"s_waitcnt\b.*lgkmcnt\(0\)", line):
                pending = False
            elif "s_barrier" in line:
                assert not pending, "s_barrier after un-waited ds_write_addtid_b32 in %s at +%d" % (name, i)
    assert checked > 0
