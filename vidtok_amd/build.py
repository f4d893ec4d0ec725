"""Build libvidtok_amd.so (gfx950) in-tree with hipcc -- no torch / cmake involved.

`python -m vidtok_amd.build` or `vidtok_amd.build.build()`; hipcc cross-compiles without a GPU.
The shared object lands next to this file (vidtok_amd/libvidtok_amd.so): git-ignored, but it
travels with a gpurun snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvidtok_amd.so")
SOURCES = ["conv_igemm_bf16.hip", "conv_igemm_f16.hip", "conv_igemm_f32.hip", "conv_igemm_x3.hip", "conv_igemm.hip", "conv_in8.hip", "conv_ws2.hip", "conv_narrow.hip", "tblock_ws128.hip", "attention.hip", "pointwise.hip", "groupnorm.hip", "regularizers.hip", "metrics.hip", "video_io.hip", "packing.hip", "error.cpp", "options.cpp", "model.cpp"]
HEADERS = ["common.h", "conv_common.h", "conv_select.h", "conv_igemm_kernel.h", "options.h", os.path.join("..", "..", "include", "vidtok_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file additions.  conv_ws2.hip: its row arithmetic runs beside the partner wave's MFMAs, where packed fp32
# (v_pk_*_f32, what the SLP vectoriser makes of eight parallel scalar chains) stalls the matrix pipe
# (profiles/r02_mfma_issue_microbench.txt) -- so no SLP there, and the instruction scheduler stays free to interleave
# the chains (the asm pins that used to keep the elements apart also kept every chain in program order: 63 s_nop of
# hazard padding in a 540-instruction row slot)
EXTRA_FLAGS = {"conv_ws2.hip": ["-fno-slp-vectorize"], "tblock_ws128.hip": ["-fno-slp-vectorize"]}


# objects whose kernels keep operands in hand-assigned registers around inline-asm MFMAs (weights in the accumulator half, results read behind
# explicit wait states): a register spill there is not a slow-down but a wrong result (round 6: the fp16 fused temporal block with 16 spilled
# registers computed garbage) -- the build refuses it.  Instantiations with cycle stamps (measurement aids, last template argument true) may spill.
NO_SPILL = {"tblock_ws128.hip": "tblock_pair_kernel", "conv_ws2.hip": "conv3x3_ws2_kernel"}
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def kernel_resources(obj):
    """[(mangled name, vgpr, agpr, spilled vgprs, scratch bytes)] of the gfx950 code object inside a compiled .o"""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        fat, dev = os.path.join(d, "fat.bin"), os.path.join(d, "dev.o")
        subprocess.check_call([f"{LLVM_BIN}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
        subprocess.check_call([f"{LLVM_BIN}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"])
        notes = subprocess.check_output([f"{LLVM_BIN}/llvm-readelf", "--notes", dev], text=True)
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)  # noqa: E731
        out.append((g("name"), int(g("vgpr_count")), int(blk.split()[0]), int(g("vgpr_spill_count")), int(g("private_segment_fixed_size"))))
    return out


def check_no_spill(objdir):
    if not all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        print("[vidtok_amd.build] spill check skipped: LLVM binutils not found under", LLVM_BIN, flush=True)
        return
    bad = []
    for src, kern in NO_SPILL.items():
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        for name, _v, _a, spill, scratch in kernel_resources(obj):
            measured = name.endswith("Lb1EEEvNS_10TBlockArgsE") or name.endswith("Lb1EEEvNS_8ConvArgsE")      # PROF = true
            if kern in name and not measured and (spill or scratch):
                bad.append(f"{name}: {spill} spilled registers, {scratch} B scratch")
    if bad:
        raise RuntimeError("register spills in hand-scheduled kernels (wrong results, not only slow):\n  " + "\n  ".join(bad))


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _src_digest(src):
    """digest of one translation unit: its source, every header (any of them may be included), its flags"""
    h = hashlib.sha256()
    for f in [src] + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(src, [])).encode())
    return h.hexdigest()


def _digest():
    h = hashlib.sha256()
    for src in SOURCES:
        h.update(_src_digest(src).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile what changed (per-object stamps under build/), link, stamp the library.  `force` recompiles everything."""
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        ostamp, odig = obj + ".stamp", _src_digest(src)
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == odig:
            continue
        if os.path.exists(ostamp):
            os.remove(ostamp)
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[vidtok_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd), ostamp, odig))
    failed = []
    for src, p, ostamp, odig in procs:
        if p.wait() != 0:
            failed.append(src)
        else:
            with open(ostamp, "w") as f:
                f.write(odig)
    if failed:
        raise RuntimeError(f"hipcc failed on {', '.join(failed)}")
    check_no_spill(objdir)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print("[vidtok_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
