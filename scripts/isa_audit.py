"""Audit the gfx950 ISA of the HIP sources for SERIALISED memory round trips.

A guarded load (`if (row < T) v = *p;`, also `ok ? *p : 0`) becomes its own basic block that ends in `s_waitcnt vmcnt(0)`,
and a load next to a store through a generic / LDS pointer is kept in program order: N such loads cost N dependent memory
round trips per wavefront instead of one.  This script compiles each source to assembly (device only, no GPU needed) and
reports, per kernel, how many `s_waitcnt vmcnt(0)` follow exactly ONE global load since the previous vector-memory wait --
the signature of that pattern.  It found the attention, LayerNorm and in-batch CE cases of round 1 (profiles/r01_gemm_pmc_sq.txt).

usage: python scripts/isa_audit.py [file.hip ...]      (default: every .hip under idvs/morec_amd/csrc)
       python scripts/isa_audit.py --seq KERNEL_SUBSTRING file.hip     print the load / wait / MFMA / store order of one kernel
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "idvs", "morec_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assembly(src):
    out = os.path.join(tempfile.gettempdir(), "isa_audit_" + os.path.basename(src) + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
           src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    cur, out = None, {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            out[cur].append(l)
        if l.startswith(".Lfunc_end"):
            cur = None
    return out


def ops_of(body):
    ops = []
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        p = t.split()
        ops.append(p[0] + (" " + p[1] if p[0] == "s_waitcnt" else ""))
    return ops


def demangle(name):
    filt = shutil.which("c++filt")
    if not filt:
        return name
    return subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()


def audit(src, threshold=3):
    rows = []
    for name, body in kernels(assembly(src)).items():
        ops = ops_of(body)
        loads = sum(o.startswith(("global_load", "buffer_load")) and "lds" not in o for o in ops)
        serial = 0
        for i, o in enumerate(ops):
            if o.startswith("s_waitcnt vmcnt(0)"):
                j, n = i - 1, 0
                while j >= 0 and not ops[j].startswith("s_waitcnt vmcnt"):
                    n += ops[j].startswith(("global_load", "buffer_load"))
                    j -= 1
                serial += n == 1
        if loads and serial >= threshold:
            rows.append((serial, loads, demangle(name)))
    return sorted(rows, reverse=True)


def sequence(src, needle):
    for name, body in kernels(assembly(src)).items():
        full = demangle(name)
        if needle not in full:
            continue
        seq = [o for o in ops_of(body) if o.startswith(("global_load", "global_store", "global_atomic", "buffer_", "v_mfma", "s_barrier",
                                                           "ds_write_b128", "ds_read_b64_tr", "scratch_")) or "vmcnt" in o]
        out, prev, c = [], None, 0
        for s in seq + [None]:
            if s == prev:
                c += 1
            else:
                if prev:
                    out.append(f"{prev} x{c}" if c > 1 else prev)
                prev, c = s, 1
        print(full[:140])
        print("  " + " | ".join(out))


def main():
    args = sys.argv[1:]
    if args and args[0] == "--seq":
        sequence(args[2], args[1])
        return
    files = args or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    for f in files:
        for serial, loads, name in audit(f):
            print(f"{os.path.basename(f)}: {serial:3d} single-load vmcnt(0) waits / {loads:3d} loads  {name[:150]}")


if __name__ == "__main__":
    main()
