"""Summarise `-Rpass-analysis=kernel-resource-usage` remarks of a build log (zstd_jni_amd.build(verbose=True) 2> log): one line per kernel."""
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name = b.split(' ')[0].split('\n')[0]
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    print("%-58s VGPR %3s AGPR %3s spillV %3s scratch %5s occ %s LDS %6s" % (name[:58], g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'), g('LDS Size [bytes/block]')))
