import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def child(level, size):
    import ctypes as C
    import __graft_entry__ as e
    from oracle import ref
    zj = e.load_package(); L = zj.lib()
    assert L.zjni_init(0) == 0
    data = zj.synth_host(size, 0, 1)
    cap = L.zjni_compressBound(len(data)); dst = C.create_string_buffer(max(cap, 1))
    t = time.time(); r = L.zjni_compress(dst, cap, data, len(data), level); dt = time.time() - t
    exp = ref.compress(data, 3, False, 14, 13) if level == 3 else ref.compress(data, level)
    print(f"  L{level} size {size} -> {r if r < 2**63 else -(2**64-r)} exp {len(exp)} {'OK' if r == len(exp) and dst.raw[:r] == exp else 'MISMATCH'} {dt*1000:.0f} ms", flush=True)
if __name__ == "__main__":
    if len(sys.argv) == 3: child(int(sys.argv[1]), int(sys.argv[2])); sys.exit(0)
    for (lvl, size) in [(1, 4096), (1, 12000), (1, 100000), (2, 131072), (3, 131072), (2, 20000)]:
        try: subprocess.run([sys.executable, __file__, str(lvl), str(size)], timeout=20)
        except subprocess.TimeoutExpired: print(f"  L{lvl} {size} TIMEOUT", flush=True)
