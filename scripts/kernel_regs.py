"""Register / spill / scratch figures of the kernels in the built scorer library
(llvm-readelf --notes on the gfx950 code object).  usage: kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.environ.get("GFALIGN_SCORER_SO") or os.path.join(ROOT, "gfalign_amd", "csrc", "libgfalign_scorer.so")
llvm = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
    subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([llvm + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    if "--isa" in sys.argv:
        out = sys.argv[sys.argv.index("--isa") + 1]
        with open(out, "w") as f:
            subprocess.check_call([llvm + "llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=f)
want = [a for a in sys.argv[1:] if not a.startswith("--") and not a.endswith(".s")]
for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    if want and not any(w in name for w in want):
        continue
    g = lambda f: re.search(r"\.%s:\s+(\d+)" % f, k).group(1)
    print("%-70s sgpr %3s spill %3s | vgpr %3s spill %3s | scratch %4s B" % (
        name[-70:], g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size")))
