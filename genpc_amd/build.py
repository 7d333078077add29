"""Builds libgenpc_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/genpc_hip.h) in-tree with hipcc.  No hipify, no torch extension machinery:
the library links only the HIP runtime and binds to whichever libamdhip64.so.7
the process already has loaded (torch's), see SURVEY.md appendix D.

    python -m genpc_amd.build [--force] [--save-temps]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgenpc_hip.so")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form",          # arithmetic is spelled out; nothing may re-fuse it
    "-fno-fast-math", "-fvisibility=hidden", "-fgpu-rdc" if False else "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "genpc_hip.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, save_temps=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    cwd = LIBDIR
    if save_temps:
        cwd = os.path.join(LIBDIR, "temps")
        os.makedirs(cwd, exist_ok=True)
        cmd.insert(1, "-save-temps")
    if verbose:
        print("[genpc_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
