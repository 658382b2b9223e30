"""Builds the native parts in-tree (no JIT cache, so the .so files travel with gpurun).

  lib/libxpbd_hip.so   HIP kernels + C ABI            (hipcc, gfx950, -ffp-contract=off)
  lib/libxpbd_host.so  C window onto the host mirror  (g++)
  lib/xpbd_headless    headless driver                (g++)
  oracle/libxpbd_oracle.so  CPU oracle, test infrastructure only (gcc)
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib")
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
ORACLE = os.path.join(ROOT, "oracle")

HIP_SOURCES = ["xpbd_kernels.hip", "xpbd_pairs.hip", "xpbd_contacts.hip", "xpbd_gjk.hip", "xpbd_world.cpp", "xpbd_multi.cpp", "xpbd_rccl.cpp"]
# -ffp-contract=off is a correctness flag: the reference (Rust) never fuses a*b+c.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall"]
CXX_FLAGS = ["-O2", "-ffp-contract=off", "-std=c++17", "-Wall", "-Wextra"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return p.stdout


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build the gfx950 kernels")
    return exe


def build_hip(force=False):
    """One object per source (only the stale ones are recompiled, a few at a time), then the link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB, exist_ok=True)
    obj_dir = os.path.join(LIB, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(LIB, "libxpbd_hip.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))] + [os.path.join(ROOT, "include", "xpbd.h")]
    jobs, objects = [], []
    for src in HIP_SOURCES:
        obj = os.path.join(obj_dir, src + ".o")
        objects.append(obj)
        if force or _newer(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([_hipcc()] + HIP_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(_run, jobs))
    if jobs or not os.path.exists(out):
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objects + ["-ldl"])
    return out


def build_host(force=False):
    hip = build_hip(force)
    deps = [os.path.join(HOST, f) for f in os.listdir(HOST)] + [hip, os.path.join(CSRC, "xpbd_math.hpp")]
    link = ["-L" + LIB, "-lxpbd_hip", "-Wl,-rpath,$ORIGIN"]
    so = os.path.join(LIB, "libxpbd_host.so")
    if force or _newer(so, deps):
        _run(["g++"] + CXX_FLAGS + ["-fPIC", "-shared", "-o", so, os.path.join(HOST, "host_capi.cpp")] + link)
    exe = os.path.join(LIB, "xpbd_headless")
    if force or _newer(exe, deps):
        _run(["g++"] + CXX_FLAGS + ["-o", exe, os.path.join(HOST, "xpbd_headless.cpp")] + link)
    return so


def build_oracle(force=False):
    so = os.path.join(ORACLE, "libxpbd_oracle.so")
    deps = [os.path.join(ORACLE, f) for f in os.listdir(ORACLE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or _newer(so, deps):
        _run(["make", "-C", ORACLE, "-B" if force else "-s"])
    return so


def build_all(force=False):
    return {"hip": build_hip(force), "host": build_host(force), "oracle": build_oracle(force)}


if __name__ == "__main__":
    import sys
    for k, v in build_all(force="--force" in sys.argv).items():
        print(k, v)
