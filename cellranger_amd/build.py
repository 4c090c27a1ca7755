"""Build libcrgpu.so (HIP, gfx950 only) in-tree with hipcc.

    python -m cellranger_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off keeps the f64 posterior
bit-identical to the reference's unfused IEEE arithmetic (lib/rust/.cargo/config.toml:5-8).
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libcrgpu.so")
OBJ_DIR = os.path.join(PKG, "_obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-value",
    "-Wno-unused-result",
] + os.environ.get("CRGPU_EXTRA_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(PKG), "include", "crgpu.h"))
    return hs


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in headers())
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > newest_header):
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("libcrgpu build failed")
    if procs or force or not os.path.exists(OUT):
        # librccl is linked directly (comm.hip: C1/C2/C3 over xGMI); torch ships a librccl.so.1 of its own, and whichever
        # of the two a process loads first serves both
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + [
            "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
