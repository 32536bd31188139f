"""Timing experiment on dw_bf16_wide (hidden 288..512: four 256 x 256 output blocks per (group, layer), every S / Z-bar strip
read by two of them): put the FOUR workgroups of one (group, layer) on the SAME XCD at the same time - linear ids 8 apart -
so that the second reader of a strip finds it in that XCD's L2 instead of HBM.   python scripts/abl_build_dwquad.py
-> experiments/abl/lib_dwquad.so (nontemporal loads kept), lib_dwquadc.so (default cache policy for the strip loads).
The source is copied to experiments/abl/src_dwquad*/ and patched there (nsfnet_amd/csrc/ is not touched); results are unchanged
(only the blockIdx -> (group, block) map and the launch grid differ).
    NSFNET_PINN_LIB=experiments/abl/lib_dwquad.so python scripts/abl_time.py --layers 8 --hidden 400 --grid 707 --what dw"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsfnet_amd import build as B


def patch(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) == count, (path, old, s.count(old))
    open(path, "w").write(s.replace(old, new))


def main():
    B.build()
    out = os.path.join(ROOT, "experiments", "abl")
    for name in ("dwquad", "dwquadc"):
        src = os.path.join(out, "src_" + name)
        shutil.rmtree(src, ignore_errors=True)
        os.makedirs(src)
        for f in os.listdir(B.CSRC):
            if f.endswith((".h", ".hip")):
                shutil.copy(os.path.join(B.CSRC, f), src)
        f = os.path.join(src, "dw_bf16_wide.hip")
        patch(f, "  const int bi = blockIdx.z / nblk, bj = blockIdx.z % nblk;\n  const int l = blockIdx.y + 1, g = blockIdx.x;\n",
              "  const bool quad = nblk == 2;      // ids 8 apart = one XCD: x = 32 (g / 8) + 8 (2 bi + bj) + g % 8\n"
              "  const int g = quad ? 8 * ((int)blockIdx.x / 32) + ((int)blockIdx.x & 7) : (int)blockIdx.x;\n"
              "  const int bi = quad ? ((int)blockIdx.x >> 4) & 1 : (int)blockIdx.z / nblk, bj = quad ? ((int)blockIdx.x >> 3) & 1 : (int)blockIdx.z % nblk;\n"
              "  const int l = blockIdx.y + 1;\n"
              "  if (g >= a.groups) return;\n")
        patch(f, "dim3(a.groups, a.L - 1, nblk * nblk), dim3(512)",
              "(nblk == 2 ? dim3(4 * ((a.groups + 7) / 8 * 8), a.L - 1, 1) : dim3(a.groups, a.L - 1, nblk * nblk)), dim3(512)")
        if name == "dwquadc":
            s = open(f).read()
            n = s.count("__builtin_nontemporal_load(")
            assert n >= 4
            open(f, "w").write(s.replace("__builtin_nontemporal_load(", "*("))
        objs = []
        for s in B.SOURCES:
            o = os.path.join(B.OBJ, s.replace(".hip", ".o"))
            if s == "dw_bf16_wide.hip":
                o = os.path.join(out, "dw_bf16_wide_%s.o" % name)
                subprocess.run([B._hipcc()] + B.FLAGS + ["-c", f, "-o", o], check=True)
            objs.append(o)
        lib = os.path.join(out, "lib_%s.so" % name)
        subprocess.run([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, check=True)
        print(lib)


if __name__ == "__main__":
    main()
