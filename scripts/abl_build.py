"""Build variant libraries for timing-only kernel experiments:  python scripts/abl_build.py NAME=FLAGS ...
e.g.  python scripts/abl_build.py noS="-DPINN_ABL=1" nothing="-DPINN_ABL=63"
Each variant recompiles the pipelined kernels with the extra flags and links them with the product build's other
objects into experiments/abl/lib_<NAME>.so (git-ignored; travels to the GPU box).  Select one at run time with
NSFNET_PINN_LIB=<path>."""
import os, subprocess, sys, shlex
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsfnet_amd import build as B

VARIANT_SOURCES = {"dw_bf16.hip", "dw_bf16_wide.hip"}      # recompiled per variant beside the pipelined / role-split sweeps


def main():
    B.build()
    out = os.path.join(ROOT, "experiments", "abl")
    os.makedirs(out, exist_ok=True)
    for spec in sys.argv[1:]:
        name, flags = spec.split("=", 1)
        objs = []
        for src in B.SOURCES:
            o = os.path.join(B.OBJ, src.replace(".hip", ".o"))
            if src in B.EXTRA_FLAGS or src in VARIANT_SOURCES:
                o = os.path.join(out, "%s_%s.o" % (src.replace(".hip", ""), name))
                cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + shlex.split(flags) + ["-c", os.path.join(B.CSRC, src), "-o", o]
                subprocess.run(cmd, check=True)
            objs.append(o)
        lib = os.path.join(out, "lib_%s.so" % name)
        subprocess.run([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, check=True)
        print(lib)

if __name__ == "__main__":
    main()
