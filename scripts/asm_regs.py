"""Register / spill summary of every kernel in a device assembly file (hipcc --cuda-device-only -S): python scripts/asm_regs.py file.s [strip-prefix]"""
import re
import sys

s = open(sys.argv[1]).read()
for blk in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)  # noqa: E731
    name = g("name")
    for pre in sys.argv[2:]:
        name = name.replace(pre, "")
    print(f"{name[:60]:60s} vgpr {g('vgpr_count'):>3s} agpr {int(blk.split()[0]):3d} spill {g('vgpr_spill_count'):>3s}")
