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
SOURCES = [os.path.join(ROOT, "radio-observer_amd", "csrc", f) for f in ("ro_kernels.hip", "ro_stft32k.hip", "ro_fourstep.hip", "ro_f64reg.hip")]


def compile_isa(dirname, sources, extra=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    text = ""
    for src in sources:
        out = os.path.join(str(dirname), os.path.basename(src) + ".s")
        cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only",
               *extra, src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text += open(out).read()
    return text


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    """the ISA of every kernel of the product library (both device sources, the flags of build.py)"""
    return compile_isa(tmp_path_factory.mktemp("isa"), SOURCES)


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
    # 8 plans x 2 sample formats x {magnitudes, spectra}, less the N = 32768 magnitude rows (stft32k_kernel), + the
    # one-kernel large transform on the N = 32768 plan x 2 formats
    assert len(stft) == 32
    assert len([k for k in ks if "stft32k_kernel" in k]) == 2
    # the four-step large transforms: the column kernel for N1 = 256, 512, 1024 x (float32 with 0, 8, 16, 32 legs nt + int16), one row kernel
    assert len([k for k in ks if "four_cols_kernel" in k]) == 15 and len([k for k in ks if "four_rows_kernel" in k]) == 1
    for name, body in ks.items():
        seen, lds = False, False
        for i, line in enumerate(body):
            if "s_barrier" in line:
                assert not seen or lds, "adjacent barriers in %s at +%d" % (name, i)
                seen, lds = True, False
            elif re.search(r"\bds_(read|write|add)|buffer_load.*\blds\b", line):
                lds = True
            elif re.match(r"\s+s_c?branch", line):
                # a branch between two barriers (stft32k_kernel's image-complete barrier sits in both arms of "has work
                # on the image or not"): what follows in the text is not what follows in time
                seen = False


def test_f64r_kernels(isa):
    """the register-resident FP64 kernel: nine plans (five of one row or sub-row per workgroup, four of 2 ... 16 rows in the
    4096-point workgroup) x three sample formats x {no gain, gain, complex spectra}; exchange 3 is lane swaps, not LDS
    (M = 8192: v_permlane16_swap, M = 16384: both); five barriers per sub-row (seven where the real and the imaginary parts
    leave through the image one after the other); 128 VGPRs at the most (16 waves per CU)"""
    ks = {k: b for k, b in kernels(isa).items() if "f64r_kernel" in k}
    assert len(ks) == 81
    spec = 0
    for name, body in ks.items():
        text = "\n".join(body)
        logm = int(re.search(r"f64r_kernelILi(\d+)E", name).group(1))
        is_spec = re.search(r"Lb([01])EEEvNS0_4ArgsE", name).group(1) == "1"
        spec += is_spec
        n32, n16 = text.count("v_permlane32_swap"), text.count("v_permlane16_swap")
        assert (n32, n16) == {12: (0, 0), 13: (0, 32), 14: (32, 32)}[logm], (name, n32, n16)
        assert text.count("s_barrier") == (7 if is_spec else 5), name
    assert spec == 27
    vg = dict(re.findall(r"\.name:\s*(\S*f64r_kernel\S*)[\s\S]*?\.vgpr_count:\s*(\d+)", isa))
    assert len(vg) == 81 and max(int(v) for v in vg.values()) <= 128, vg


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


# ---------------------------------------------------------------------------
# Hazards that hipcc does not pad for us
# ---------------------------------------------------------------------------
def _regs(tok):
    """VGPR numbers named by an operand like v12 or v[12:15]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _instr(line):
    """(mnemonic, operand list) of an instruction line, or None for labels / comments / directives"""
    m = re.match(r"^\s+([a-z][a-z0-9_]*)\s*(.*)$", line)
    if not m or line.lstrip().startswith((";", ".")):
        return None
    ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", m.group(2).split(";")[0].strip()) if t.strip()]
    return m.group(1), ops


def _valu_defs(mn, ops):
    """VGPRs a VALU instruction writes"""
    if not mn.startswith("v_") or not ops:
        return set()
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    d = _regs(ops[0])
    if mn.startswith(("v_permlane32_swap", "v_permlane16_swap", "v_swap")) and len(ops) > 1:
        d |= _regs(ops[1])
    return d


def wide_store_hazards(text):
    """every 12/16-byte store whose data registers a VALU instruction overwrites within the next two wait states
    (ro_device_util.h, buf_store_f4): [(kernel, line offset, store, writer)]"""
    bad, stores = [], 0
    for name, body in kernels(text).items():
        for i, line in enumerate(body):
            ins = _instr(line)
            if not ins or not re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]$", ins[0]):
                continue
            stores += 1
            data = _regs(ins[1][1] if ins[0].startswith(("global", "flat", "scratch")) else ins[1][0])
            assert data, line
            waited, j = 0, i + 1
            while waited < 2 and j < len(body):
                nxt = _instr(body[j])
                j += 1
                if not nxt:
                    continue
                if _valu_defs(*nxt) & data:
                    bad.append((name, i, line.strip(), body[j - 1].strip()))
                    break
                waited += int(nxt[1][0], 0) + 1 if nxt[0] == "s_nop" else 1
    return bad, stores


def test_no_valu_write_to_store_data_within_two_wait_states(isa):
    bad, stores = wide_store_hazards(isa)
    assert stores > 100           # the row stores of every plan, fold / interleave / FP64 / chirp-z kernels
    assert not bad, bad[:5]


def test_the_store_hazard_check_sees_an_unguarded_build(tmp_path):
    """the same kernels without buf_store_f4's wait states: the check above must go red on them, or it proves nothing"""
    text = compile_isa(tmp_path, SOURCES[1:], extra=("-DRO_STORE_NOP=0",))
    bad, stores = wide_store_hazards(text)
    assert stores >= 16 and bad, "no hazard found in the unguarded build of stft32k_kernel (%d stores)" % stores


def test_square_roots_are_not_stored_to_lds_straight_away(isa):
    """v_sqrt_f32 runs in the transcendental pipe and the add-TID writes that take its result sit in inline asm, where
    hipcc pads nothing: the kernels keep at least four wait states between the two (image writes one pair late)."""
    checked = 0
    for name, body in kernels(isa).items():
        last_sqrt = {}                      # VGPR -> wait states since the v_sqrt_f32 that wrote it
        for line in body:
            ins = _instr(line)
            if not ins:
                continue
            mn, ops = ins
            step = int(ops[0], 0) + 1 if mn == "s_nop" else 1
            if mn == "ds_write_addtid_b32":
                for r in _regs(ops[0].split()[0]):
                    if r in last_sqrt:
                        checked += 1
                        assert last_sqrt[r] >= 4, "%s: v%d stored %d wait states after its v_sqrt_f32" % (name, r, last_sqrt[r])
            for r in list(last_sqrt):
                last_sqrt[r] += step
            for r in _valu_defs(mn, ops):
                last_sqrt.pop(r, None)
            if mn.startswith("v_sqrt_f32"):
                for r in _regs(ops[0]):
                    last_sqrt[r] = 0
    assert checked >= 64


def test_row_priority_is_set_where_workgroups_share_a_cu(isa):
    """stft_kernel raises and lowers its priority by rows done in the plans of N >= 1024; the one-wave
    plans below and the one-workgroup-per-CU kernel of N = 32768 do not touch it."""
    with_prio, without = 0, 0
    for name, body in kernels(isa).items():
        if "stft_kernel" not in name:
            continue
        m = re.search(r"PlanILi(\d+)E", name)
        assert m, name
        n = int(m.group(1))
        levels = {l.split()[-1] for l in body if re.match(r"\s+s_setprio\b", l)}
        if n >= 1024:
            assert levels == {"0", "1", "2", "3"}, (name, levels)
            with_prio += 1
        else:
            assert not levels, (name, levels)
            without += 1
    assert with_prio >= 20 and without == 8
    for name, body in kernels(isa).items():
        if "stft32k_kernel" in name:
            assert not any(re.match(r"\s+s_setprio\b", l) for l in body), name
