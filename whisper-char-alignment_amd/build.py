"""Builds libwca.so (hand-written HIP for gfx950 + the C ABI of include/wca.h) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container too.
Usage:  python whisper-char-alignment_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libwca.so")
SOURCES = ["gemm.hip", "gemm_rows.hip", "attention.hip", "attention_split.hip", "elementwise.hip", "logmel.hip", "postproc.hip", "dtw.hip", "decode.hip", "engine.hip", "flac.cpp", "debug_switch.cpp"]
HEADERS = ["kernels.h", "wca_common.h", "gemm_epilogue.h", os.path.join("..", "..", "include", "wca.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (the softmax / epilogue VALU code reads them
# directly; the AGPR form costs a v_accvgpr_read/write pair per element in the attention loop)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form"]
# attention.hip: no NaN can occur in the online softmax (masked scores are -inf, the -inf - -inf cases are guarded), and
# without this flag every fmaxf on a cross-lane / MFMA result is preceded by a canonicalising v_max_f32 x, x
# (16 of the 24 v_max per 64-key tile, in a VALU-bound loop)
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans"], "attention_split.hip": ["-fno-honor-nans"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        if s.endswith(".cpp"):  # host-only source (audio file decoding): plain C++, no device code
            cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-x", "c++", "-c", s, "-o", o]
        else:
            cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose:
            print("[wca build] compiled", os.path.basename(s), flush=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[wca build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
