import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import precision_probe as pp
for shape in [(6, 256, 4000), (6, 128, 4000), (4, 50, 4000)]:
    for m in [("bf16x3","bf16x3","bf16"), ("bf16x3","bf16","bf16x3"), ("bf16","bf16x3","bf16x3")]:
        pp.run(*shape, m)
