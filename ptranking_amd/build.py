"""Builds ptranking_amd/libptranking_amd.so (hand-written HIP kernels + the C ABI) for gfx950 with hipcc.

`python -m ptranking_amd.build [--force]`.  hipcc cross-compiles without a GPU; the resulting .so is git-ignored
but travels with the tree to the GPU box.  One translation unit per kernel family, linked into one shared library
whose only runtime dependency is the HIP runtime (libamdhip64) — no torch types anywhere in the ABI.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "csrc", "build")
LIB_PATH = os.path.join(PKG_DIR, "libptranking_amd.so")
ARCH = "gfx950"
SOURCES = ["abi.hip", "pairwise.hip", "pairwise_ring.hip", "lambdaloss.hip", "approxndcg.hip", "listwise.hip", "metrics.hip", "scorer.hip", "scorer_bwd.hip", "scorer_x6.hip", "scorer_bwd_x6.hip", "scorer_dw_x6.hip", "linear.hip", "linear_x6.hip", "linear_bw_x6.hip", "bnact.hip", "listsf.hip", "train_step.hip", "letor.cpp"]
HEADERS = ["ptr_device.h", "ptr_dropout.h", "ptr_mlp.h", "ptr_ring.h", "ptr_linear.h", os.path.join("..", "..", "include", "ptranking_amd.h")]
CXXFLAGS = ["-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-ffp-contract=off"]


# per-source flags.  r6 measured `-mllvm -amdgpu-sched-strategy=max-ilp` on the LambdaRank ring kernel (csrc/pairwise_ring.hip): the ILP strategy removes
# the 59 s_nop wait states per ring step the default strategy leaves behind the transcendentals (562 -> 501 issue slots per 32 pair evaluations at 256
# documents) but needs 208 instead of 150 registers — two waves per SIMD instead of three — and the kernel got SLOWER (92.9 -> 101.7 us at 4096 x 256);
# capped at three waves per SIMD it emits 140 wait states.  Not used; the hook stays for the next such experiment.
EXTRA_FLAGS = {}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(tag, defines, sources=("scorer_bwd.hip",), verbose=False):
    """Experiment helper: libptranking_amd.<tag>.so = the current objects with `sources` recompiled under extra -D defines.
    Selected at run time with PTR_LIB=<path> (see _lib.py).  Not used by the product path."""
    hipcc = _hipcc()
    build()
    vdir = os.path.join(OBJ_DIR, tag)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for s in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
        if s in sources:
            obj = os.path.join(vdir, os.path.splitext(s)[0] + ".o")
            cmd = [hipcc] + CXXFLAGS + EXTRA_FLAGS.get(s, []) + [d if d.startswith("-") else f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, s), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(obj)
    out = os.path.join(PKG_DIR, f"libptranking_amd.{tag}.so")
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs, check=True)
    return out


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library.  Returns the library path."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            flags = CXXFLAGS if s.endswith(".hip") else [f for f in CXXFLAGS if not f.startswith("--offload-arch")
                                                              and f != "-fno-gpu-rdc"] + ["-pthread"]
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # python -m ptranking_amd.build --variant TAG DEF[=V] ... [--src a.hip,b.hip]
        args = sys.argv[3:]
        srcs = ("scorer_bwd.hip",)
        if "--src" in args:
            i = args.index("--src")
            srcs = tuple(args[i + 1].split(","))
            args = args[:i] + args[i + 2:]
        print(build_variant(sys.argv[2], args, srcs, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
