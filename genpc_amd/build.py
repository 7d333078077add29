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
# No packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) anywhere in the library: in the farthest-point
# sampling they were what drew wrong samples beside other streams' matrix instructions (csrc/fps.hip; the low half
# of a register pair, lanes 48-63); scalar fp32 instructions give the same bits.  GENPC_PACKED_FP32=1 builds with them (A/B).
# (The feature is the device compiler's: the host pass prints one "not a recognized feature" line per file, dropped below.)
if os.environ.get("GENPC_PACKED_FP32", "0") != "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "genpc_hip.h"))
    hs.append(os.path.abspath(__file__))
    return hs


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force=False, save_temps=False, verbose=True, jobs=None):
    """One object per .hip file (stale ones only, compiled side by side), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    cflags = [f for f in FLAGS if f != "-shared"]
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [HIPCC] + cflags + ["-c", src, "-o", obj]
            cwd = objdir
            if save_temps:
                cwd = os.path.join(LIBDIR, "temps")
                os.makedirs(cwd, exist_ok=True)
                cmd.insert(1, "-save-temps")
            todo.append((cmd, cwd))

    def run(job):
        if verbose:
            print("[genpc_amd.build]", " ".join(job[0]), flush=True)
        r = subprocess.run(job[0], cwd=job[1], stderr=subprocess.PIPE, text=True)
        noise = "'-packed-fp32-ops' is not a recognized feature for this target"
        err = "\n".join(l for l in r.stderr.splitlines() if noise not in l)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, job[0])

    jobs = jobs or int(os.environ.get("GENPC_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(run, todo))
    link = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", LIB]
    if verbose:
        print("[genpc_amd.build]", " ".join(link), flush=True)
    subprocess.check_call(link, cwd=LIBDIR)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
