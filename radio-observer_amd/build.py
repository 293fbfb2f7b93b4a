"""Build the gfx950 shared library (libro_stft.so) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as
well as on the MI355X box.  The .so is git-ignored but travels with gpurun.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libro_stft.so")
SOURCES = ["ro_kernels.hip", "ro_stft32k.hip", "ro_fourstep.hip", "ro_f64reg.hip", "ro_stft_capi.cpp", "ro_abi_helpers.cpp",
           "ro_exchange.cpp", "ro_stream.cpp", "ro_czt.cpp"]
HEADERS = ["ro_kernels.h", "ro_host.h", "ro_fft_device.h", "ro_fft_planar.h", "ro_k32_lds.h", "ro_device_util.h", "ro_f64_device.h", "ro_narrow.h", os.path.join("..", "..", "include", "ro_stft.h")]

# -fno-slp-vectorize: the SLP vectoriser turns the twiddle multiplies into v_pk_* ops whose
# constant operands must sit in VGPR pairs; that costs ~28 VGPRs and makes the 1024-thread
# N=32768 kernel spill (see DESIGN.md "register budget").
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
