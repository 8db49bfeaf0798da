"""Static reports on the device assembly hipcc emits for the kernels in ptranking_amd/csrc (CPU only: cross-compiles for gfx950).

    python scratch/isa_report.py deps   [file.hip ...]     back-to-back DEPENDENT matrix instructions per kernel (a dependent
                                                           v_mfma_f32_16x16x4_f32 issues after 40 cycles, an independent one after 32)
    python scratch/isa_report.py blocks file.hip KERNEL    per basic block with >= 20 MFMAs: MFMA / VALU / LDS / VMEM / SALU counts and the VALU mix
    python scratch/isa_report.py regs   [file.hip ...]     VGPRs, spills and scratch bytes per kernel

What found the forward's dependent chains (r3: 352 -> 340 us) and the layer-wise dZ kernel's (147 of 196 MFMAs)."""
import os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ptranking_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++20", "-fno-gpu-rdc", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]


sys.path.insert(0, ROOT)
from ptranking_amd.build import EXTRA_FLAGS       # per-source flags of the product build (e.g. the ring kernel's scheduling strategy)


def asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run([HIPCC, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    """{mangled name: [instruction lines]}"""
    res, name, cur = {}, None, []
    for l in text.splitlines():
        m = re.match(r"^(_Z\S+|[a-z_][a-z_0-9]*):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            name, cur = m.group(1), []
            continue
        t = l.split(";")[0].rstrip()
        if name is None or not t.strip():
            continue
        cur.append(t.strip())
        if t.strip().startswith("s_endpgm"):
            res[name] = cur
            name = None
    return res


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return n


def deps(files):
    for f in files:
        for name, k in kernels(asm(f)).items():
            prev, n, dep = None, 0, 0
            for t in k:
                if t.startswith("v_mfma"):
                    m = re.match(r"v_mfma_\S+ (\S+), (\S+), (\S+), (\S+)", t)
                    n += 1
                    dep += prev == m.group(4)
                    prev = m.group(1)
            if n:
                print(f"{os.path.basename(f):16s} {demangle(name)[:90]:90s} MFMAs {n:5d}  dependent on the one before {dep:5d}")


def blocks(f, pat):
    for name, k in kernels(asm(f)).items():
        if pat not in name and pat not in demangle(name):
            continue
        print(demangle(name))
        cur, label = [], "entry"
        for t in k + ["end:"]:
            if t.endswith(":"):
                mf = sum(x.startswith("v_mfma") for x in cur)
                if mf >= 20:
                    v = [x.split()[0] for x in cur if x.startswith("v_") and not x.startswith("v_mfma")]
                    print(f"  {label:14s} instr {len(cur):4d} mfma {mf:3d} valu {len(v):3d} lds {sum(x.startswith('ds_') for x in cur):2d} "
                          f"vmem {sum(x.startswith(('global_', 'scratch_', 'buffer_')) for x in cur):2d} salu {sum(x.startswith('s_') for x in cur):3d}  "
                          + ", ".join(f"{a} {b}" for a, b in Counter(v).most_common(8)))
                cur, label = [], t[:-1]
            else:
                cur.append(t)


def regs(files):
    for f in files:
        r = subprocess.run([HIPCC, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(f), []), "-c", f, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        name, row = None, {}
        for l in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", l)
            if m:
                name = m.group(1)
            m = re.search(r" (VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", l)
            if m and name:
                row.setdefault(name, {})[m.group(1)] = m.group(2)
        for n, v in row.items():
            print(f"{os.path.basename(f):16s} {demangle(n)[:90]:90s} VGPRs {v.get('VGPRs'):>4s} spill {v.get('VGPRs Spill'):>4s} scratch {v.get('ScratchSize [bytes/lane]'):>4s}")


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "deps"
    files = [a if os.path.exists(a) else os.path.join(CSRC, a) for a in sys.argv[2:] if a.endswith(".hip")]
    allf = [os.path.join(CSRC, x) for x in ("scorer.hip", "scorer_bwd.hip", "linear.hip", "listsf.hip")]
    if cmd == "deps":
        deps(files or allf)
    elif cmd == "regs":
        regs(files or allf)
    elif cmd == "blocks":
        blocks(files[0], sys.argv[3])
    else:
        raise SystemExit(__doc__)
