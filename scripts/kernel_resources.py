"""VGPR / AGPR / SGPR / LDS / scratch of every kernel in the HIP sources (device-only compile to assembly, no GPU needed) and whether a
wave of it fits NEXT TO the persistent eight-phase GEMMs on a SIMD (512 VGPR lanes-registers per SIMD: two gemm8p waves of 224 leave 64;
160 KiB of LDS per CU: their 128-KiB K-tile buffers + epilogue slices leave < 32 KiB).   python scripts/kernel_resources.py [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "idvs", "morec_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
for src in files:
    out = os.path.join(tempfile.gettempdir(), "kres_" + os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                    "--cuda-device-only", src, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        def f(k):
            r = re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body)
            return int(r.group(1)) if r else 0
        nv, acc_off, lds, scr = f("next_free_vgpr"), f("accum_offset"), f("group_segment_fixed_size"), f("private_segment_fixed_size")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0][:70]
        alloc = (nv + 7) // 8 * 8
        print(f"{os.path.basename(src):22s} {dem:70s} vgpr+agpr {nv:3d} (alloc {alloc:3d}; arch {acc_off:3d})  lds {lds:6d}  scratch {scr:4d}  {'fits beside gemm8p' if alloc <= 64 and lds <= 24 * 1024 else ''}")
