"""Builds libvlp_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m vlp_amd.build            # incremental
    python -m vlp_amd.build --force

The library has no torch / python dependency; it is loaded with ctypes (vlp_amd/_lib.py).  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
BUILD = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libvlp_hip.so")
SOURCES = ["api.cpp", "gemm_nt.hip", "gemm_nt_wp.hip", "gemm_nt_ps.hip", "gemm_nt_splitk.hip", "gemm_tn.hip", "attention.hip", "layernorm.hip", "elementwise.hip", "loss.hip", "adam.hip", "pretext.hip", "decode.hip"]
# investigation variants (phased / k32 NT kernels, further wave-pipelined configurations, two-kernel attention backward, stream-K grouped
# wgrad): `python -m vlp_amd.build --lab` -> vlp_amd/libvlp_hip_lab.so (-DVLP_LAB_BUILD), selected with VLP_HIP_LIB=...; never the product library
LAB_SOURCES = ["gemm_nt_ph.hip", "gemm_nt_k32.hip"]
LAB_LIB = os.path.join(HERE, "libvlp_hip_lab.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-Wno-unused-result", "-ffp-contract=fast"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, lab=False):
    if lab:
        return _build(force, verbose, os.path.join(CSRC, "build_lab"), LAB_LIB, SOURCES + LAB_SOURCES, ["-DVLP_LAB_BUILD"])
    return _build(force, verbose, BUILD, LIB, SOURCES, [])


def _build(force, verbose, BUILD, LIB, SOURCES, extra):
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(INCLUDE, "vlp_hip.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(BUILD, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [HIPCC] + FLAGS + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, rc, out in ex.map(run, jobs):
                if verbose and out.strip():
                    print("\n".join(l for l in out.splitlines() if "warning" not in l and "note:" not in l and "|" not in l and "^" not in l), file=sys.stderr)
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s" % src)
                if verbose:
                    print("compiled %s" % src)
    if jobs or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("linked %s" % LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, lab="--lab" in sys.argv)
