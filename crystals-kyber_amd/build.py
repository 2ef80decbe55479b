"""Build the native libraries IN-TREE (they travel to the GPU box with gpurun; they are git-ignored).

  crystals-kyber_amd/libmlkem_amd.so  hipcc --offload-arch=gfx950 : kernels + batch C-ABI (include/mlkem_batch.h)
  crystals-kyber_amd/libml_kem.so     gcc : ml_kem.h-compatible drop-in shim, linked against libmlkem_amd.so
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmlkem_amd.so")
SHIM = os.path.join(HERE, "libml_kem.so")


# -fno-slp-vectorize: left to itself the compiler packs pairs of the kernels' scalar fp32 operations into v_pk_fma_f32 /
# v_pk_add_f32, which issue at half rate on gfx950 (no gain) and cost moves + registers to form the operand pairs: the K-PKE
# kernels are 3-5 % faster without it (profiles/r03_kpke_experiments.txt).
# -fvisibility=hidden: the library exports the C-ABI of include/mlkem_batch.h (MLKEM_API) and nothing else.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fvisibility=hidden"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def hipcc_path():
    for p in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    raise RuntimeError("hipcc not found")


def build(force=False):
    root = os.path.dirname(HERE)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(root, "include", "mlkem_batch.h"),
                                                                 os.path.join(root, "include", "mlkem_compat.h")]
    if force or not _newer(LIB, srcs):
        _run([hipcc_path(), *HIPCC_FLAGS, "-fPIC", "-shared", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB,
              os.path.join(CSRC, "mlkem_capi.hip")])
    if force or not _newer(SHIM, srcs + [LIB]):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-o", SHIM, os.path.join(CSRC, "ml_kem_shim.c"),
              "-L" + HERE, "-lmlkem_amd", "-Wl,-rpath,$ORIGIN"])
    return LIB, SHIM


if __name__ == "__main__":
    print(build(force=True))
