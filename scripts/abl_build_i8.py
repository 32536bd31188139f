"""Timing-only libraries that price int8-limb MFMAs INSIDE the product kernels before
anything real is built:   python scripts/abl_build_i8.py   ->  experiments/abl/lib_i8bwd.so, lib_i8dw.so

The sources are copied to experiments/abl/src_i8/ and patched THERE (nsfnet_amd/csrc/ is not touched):
 * MFMA_Q(q, a, b, c) becomes ONE v_mfma_i32_32x32x32_i8 on the same operand registers for even k-steps and nothing for
   odd ones: per two bf16 k-steps (K = 32) three i8 MFMAs instead of six bf16 ones - the instruction count, operand
   registers and LDS reads of a 15-bit x 15-bit limb product (a1 w1, a1 w0, a0 w1).  Results are wrong on purpose.
 * bwd_split: where an E phase reads an accumulator element it converts it i32 -> f32 and scales it (v_cvt + v_mul per
   element, what the real thing needs too), so that the E phases downstream keep seeing finite O(1) pseudo-random data
   (NaN or zero data would lower the power and flatter the variant: profiles/r03_ablations.txt D).
 * dw_bf16: the software pipeline's interleave is re-paced for half the MFMAs (the conversion VALU stays).
NOT modelled in i8bwd / i8dw: the extra VALU of quantising to limbs (row maxima, scaling, byte packing) in the E phases /
the staging.  i8dwq = i8dw plus 4 VALU per staged value in dW (what scaling, rounding, limb split and byte packing add to
the bf16 split that stays in place as the stand-in of the limb-plane writes).
Each library patches ONE kernel, so the other kernels see real data:  bash scripts/run_abl.sh bwd,dw base i8bwd i8dw i8dwq"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsfnet_amd import build as B

I8 = '''
__device__ __forceinline__ f32x16 mfma_i8_standin(u32x4 a, u32x4 b, f32x16 c) {
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  typedef int i32x16v __attribute__((ext_vector_type(16)));
  i32x16v ci = __builtin_bit_cast(i32x16v, c);
  ci = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a), __builtin_bit_cast(i32x4v, b), ci, 0, 0, 0);
  return __builtin_bit_cast(f32x16, ci);
}
#define MFMA_Q(q, a, b, c) ((((q) & 1)) ? (c) : mfma_i8_standin(a, b, c))
'''
CONVERT = "  return (float)__builtin_bit_cast(int, acc_elem) * 2e-6f;\n#else"


# +4 VALU per staged value (32 values per thread and chunk): scale, round to integer, split into two limbs, pack the bytes -
# against the ~2.5 per value of the bf16 hi / lo split that stays in place (its image writes are the limb planes' stand-in)
QUANT_VALU = '''
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        asm volatile("v_fma_f32 %0, %0, 1.0, 0\\n\\tv_fma_f32 %0, %0, 1.0, 0\\n\\tv_fma_f32 %0, %0, 1.0, 0\\n\\tv_fma_f32 %0, %0, 1.0, 0" : "+v"(zr[s][e]));
        asm volatile("v_fma_f32 %0, %0, 1.0, 0\\n\\tv_fma_f32 %0, %0, 1.0, 0\\n\\tv_fma_f32 %0, %0, 1.0, 0\\n\\tv_fma_f32 %0, %0, 1.0, 0" : "+v"(av[s][e]));
      }
'''


def patch(path, old, new):
    s = open(path).read()
    assert s.count(old) == 1, (path, old)
    open(path, "w").write(s.replace(old, new))


def main():
    B.build()
    out = os.path.join(ROOT, "experiments", "abl")
    src = os.path.join(out, "src_i8")
    shutil.rmtree(src, ignore_errors=True)
    os.makedirs(src)
    for f in os.listdir(B.CSRC):
        if f.endswith((".h", ".hip")):
            shutil.copy(os.path.join(B.CSRC, f), src)
    patch(os.path.join(src, "bf16_util.h"),
          "#define MFMA_Q(q, a, b, c) (PINN_ABL_SHAPE16 ? mfma_bf16_shape16((q) & 1, a, b, c) : mfma_bf16(a, b, c))", I8)
    patch(os.path.join(src, "bwd_bf16_split.hip"), "  return acc_elem;\n#else", CONVERT)
    patch(os.path.join(src, "dw_bf16.hip"), "constexpr int NMF = (DI::CH / 16) * TM * TN * (TERMS == 3 ? 3 : 1);",
          "constexpr int NMF = (DI::CH / 16) * TM * TN * (TERMS == 3 ? 3 : 1) / 2;")
    patch(os.path.join(src, "dw_bf16.hip"), "P24 ? 5 : 4, 0);", "P24 ? 10 : 8, 0);")
    srcq = os.path.join(out, "src_i8q")
    shutil.rmtree(srcq, ignore_errors=True)
    shutil.copytree(src, srcq)
    patch(os.path.join(srcq, "dw_bf16.hip"), "    unsigned char* base = ldsb + (size_t)buf * 4 * DI::ARR + p * DI::RSB + (og ^ (p & 6)) * 8;", QUANT_VALU +
          "    unsigned char* base = ldsb + (size_t)buf * 4 * DI::ARR + p * DI::RSB + (og ^ (p & 6)) * 8;")
    patch(os.path.join(srcq, "dw_bf16.hip"), "P24 ? 10 : 8, 0);", "P24 ? 15 : 13, 0);")
    for name, victim in (("i8bwd", "bwd_bf16_split.hip"), ("i8dw", "dw_bf16.hip"), ("i8dwq", "dw_bf16.hip")):
        if name == "i8dwq":
            src = srcq
        objs = []
        for s in B.SOURCES:
            o = os.path.join(B.OBJ, s.replace(".hip", ".o"))
            if s == victim:
                o = os.path.join(out, "%s_%s.o" % (s.replace(".hip", ""), name))
                cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(s, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(src, s), "-o", o]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise SystemExit(r.stderr[-4000:])
                spills = [l for l in r.stderr.splitlines() if "VGPRs Spill" in l and ": 0 [" not in l]
                print("%s: %d kernels with VGPR spills" % (name, len(spills)))
            objs.append(o)
        lib = os.path.join(out, "lib_%s.so" % name)
        subprocess.run([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, check=True)
        print(lib)


if __name__ == "__main__":
    main()
