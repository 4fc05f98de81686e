"""Build libmsn_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m multimodal_supernovae_amd.build [--force]

Each csrc/*.hip is compiled to build/<name>.o (only when older than its sources) and the
objects are linked into multimodal_supernovae_amd/lib/libmsn_hip.so.  hipcc cross-compiles
without a GPU; the .so is git-ignored but travels with a gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmsn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "msn_hip.h"))
    if force or _newer(obj, [src] + headers):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        # every kernel a translation unit launches must have been instantiated by one of them: resolve all symbols now (hipcc's host
        # pass has dropped explicit instantiations without a diagnostic -- the library linked and failed only when it was loaded)
        import ctypes
        try:
            ctypes.CDLL(LIB, mode=os.RTLD_NOW)
        except OSError as e:
            raise RuntimeError(f"{LIB} does not load: {e}") from e
        if verbose:
            print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} objects")
    elif verbose:
        print(f"{LIB} is up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
