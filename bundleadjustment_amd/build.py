"""In-tree build of the native libraries (no cmake: two compiler invocations).

    python -m bundleadjustment_amd.build [--force]

libba_host.so : g++   bundleadjustment_amd/host/ba_host.cpp
libba_gpu.so  : hipcc --offload-arch=gfx950 bundleadjustment_amd/csrc/*.hip  (cross-compiles without a GPU; one object per
                source under bundleadjustment_amd/build/, so a change recompiles one file)
ba_main       : g++   bundleadjustment_amd/host/ba_main.cpp, linked against both
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
INC = os.path.join(ROOT, "include")
HOST_SO = os.path.join(PKG, "libba_host.so")
GPU_SO = os.path.join(PKG, "libba_gpu.so")
MAIN_BIN = os.path.join(PKG, "ba_main")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_host(force=False):
    src = [os.path.join(PKG, "host", "ba_host.cpp")]
    deps = src + glob.glob(os.path.join(INC, "*.h"))
    if force or _stale(HOST_SO, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", INC] + src + ["-o", HOST_SO])
    return HOST_SO


def build_gpu(force=False):
    """One object per source under build/ (recompiled when the source or a header is newer), in parallel, then one link."""
    from concurrent.futures import ThreadPoolExecutor
    src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
    hdr = glob.glob(os.path.join(PKG, "csrc", "*.h")) + glob.glob(os.path.join(INC, "*.h"))
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
             "-I", INC, "-I", os.path.join(PKG, "csrc")]
    objs, todo = [], []
    for f in src:
        o = os.path.join(objdir, os.path.basename(f)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [f] + hdr):
            todo.append([hipcc] + flags + ["-c", f, "-o", o])
    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(_run, todo))
    if todo or force or _stale(GPU_SO, objs):
        _run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs +
             ["-L", os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", GPU_SO])
    return GPU_SO


def build_main(force=False):
    src = [os.path.join(PKG, "host", "ba_main.cpp")]
    if not os.path.exists(src[0]):
        return None
    if force or _stale(MAIN_BIN, src + [HOST_SO, GPU_SO]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-I", INC] + src +
             ["-L", PKG, "-lba_host", "-lba_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib"),
              "-o", MAIN_BIN])
    return MAIN_BIN


def build_all(force=False):
    build_host(force)
    build_gpu(force)
    build_main(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
