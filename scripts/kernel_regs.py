"""Register / scratch usage of every kernel in a built object of vidtok_amd/build (build container, no GPU):
    python scripts/kernel_regs.py conv_igemm [filter]
unbundles the gfx950 code object from the .hip_fatbin section and prints vgpr / agpr / sgpr counts, spills and scratch
bytes per kernel -- a spill in a K loop or a second register-allocation granule (> 256) shows up here before any run."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "conv_igemm"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    obj = os.path.join(ROOT, "vidtok_amd", "build", name + ".o")
    with tempfile.TemporaryDirectory() as d:
        fat, dev = os.path.join(d, "fat.bin"), os.path.join(d, "dev.o")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", dev], text=True)
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)  # noqa: E731
        try:
            dem = subprocess.check_output([f"{LLVM}/llvm-cxxfilt", g("name")], text=True).strip()
        except Exception:
            dem = g("name")
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
        if flt and flt not in dem:
            continue
        print(f"vgpr {g('vgpr_count'):>3s} agpr {int(blk.split()[0]):3d} sgpr {g('sgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} "
              f"scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}  {dem[:150]}")


if __name__ == "__main__":
    main()
