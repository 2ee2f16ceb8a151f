"""Register / scratch summary of every kernel in a hipcc -S listing:  python tools/isa_stats.py file.s"""
import re, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk)
    name = g("name").group(1)
    print(f"{name[:70]:70s} vgpr {g('vgpr_count').group(1):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count').group(1):>4s} "
          f"spill {g('vgpr_spill_count').group(1):>4s} scratch {g('private_segment_fixed_size').group(1):>5s}")
