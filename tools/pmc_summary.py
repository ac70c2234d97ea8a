"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/pmc2/l<level>_<counter>/, one pass per counter,
driver = tools/prof_driver.py 65536 65536 <level> 1) into profiles/r01_pmc_traffic.json, which bench.py reads for
roofline.traffic.  Counter units are KB; calibration of what one request tallies: tools/micro/chase cal."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc2")
n, size = 65536, 65536
out = {"note": "HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes "
               "(TCC_EA0_RDREQ/WRREQ based). Calibrated on this kernel's access pattern with tools/micro/chase: one random "
               "4-byte read miss tallies 63.7 B, one random 4-byte write tallies 32 B; the x2 correction the guide gives "
               "for wide streaming reads does not apply to these scattered 4-8 byte accesses, so values are uncorrected.",
       "calibration": {"random_4B_reads_per_launch": 131072000, "FETCH_SIZE_KB": 8149800.7, "WRITE_SIZE_KB_readwrite_launch": 4103834.5}}
for level in (1, 3):
    rec = collections.defaultdict(lambda: {"fetch": [], "write": []})
    for cn, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        f = glob.glob(os.path.join(src, f"l{level}_{cn}", "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            name = r["Kernel_Name"].split("(")[0]
            if name.startswith("zj_"):
                rec[name][key].append(float(r["Counter_Value"]) * 1024.0)
    summ = {}
    for name, v in rec.items():
        # kernels launched several times per call (list A / list B / sweep): keep the per-call sum of the largest launches
        fetch = max(v["fetch"]) if v["fetch"] else 0.0
        write = max(v["write"]) if v["write"] else 0.0
        summ[name] = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    out[f"L{level}_{n}x{size}"] = summ
with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps({k: {kk: round(vv["hbm_bytes_per_launch"] / 1e9, 2) for kk, vv in v.items()} for k, v in out.items() if k.startswith("L")}, indent=1))
