"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/measure_round.sh: gpurun_out/<tag>/pmc/<config>_L<level>_<n>x<size>_<COUNTER>.csv,
one counter per pass, driver = tools/prof_driver.py <n> <size> <level> 1; tools/pmc_traffic.sh does the same and keeps the driver's JSON line) into
profiles/<round>_pmc_traffic.json, which bench.py reads for roofline.traffic.  Every record carries the zjni_build_stamp() the driver printed
(revision + hash of csrc/): bench.py quotes a figure only when its own library has that stamp.  Counter units are KB; what one request tallies: tools/micro/chase cal.
usage: pmc_summary.py <gpurun_out/tag> <round, e.g. r02>   (copies the csvs to profiles/<round>_pmc/ too)"""
import collections, csv, datetime, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "zstd-jni_amd/csrc"], text=True).strip())
out = {"note": "HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes "
               "(TCC_EA0_RDREQ/WRREQ based). Calibrated on this kernel's access pattern with tools/micro/chase: one random "
               "4-byte read miss tallies 63.7 B, one random 4-byte write tallies 32 B; the x2 correction the guide gives "
               "for wide streaming reads does not apply to these scattered 4-8 byte accesses, so values are uncorrected. "
               "Counter collection serialises kernels: the level-3 'metric' passes run with ZJNI_NEED_INLINE=1 (flag kernel ahead of the match kernel on one "
               "stream, so every picked frame has its flags from its first round; in production they arrive beside the match kernel during its first ~30 ms "
               "of ~150), and 'metricnoflags' (ZJNI_NEED=0) is the same launch with no flags at all: real traffic lies between the two, close to the first.",
       "calibration": {"random_4B_reads_per_launch": 131072000, "FETCH_SIZE_KB": 8149800.7, "WRITE_SIZE_KB_readwrite_launch": 4103834.5},
       "measured_on_commit": head + ("+uncommitted csrc changes" if dirty else ""), "measured_on_date": datetime.date.today().isoformat(),
       "driver": "tools/pmc_traffic.sh -> tools/prof_driver.py <n> <size> <level> 1"}
dst = os.path.join(ROOT, "profiles", rnd + "_pmc"); os.makedirs(dst, exist_ok=True)
keys = sorted({re.sub(r"_(FETCH|WRITE)_SIZE\.csv$", "", os.path.basename(p)) for p in glob.glob(os.path.join(src, "pmc", "*_SIZE.csv"))})
for key in keys:
    rec = collections.defaultdict(lambda: {"fetch": [], "write": []})
    for cn, k in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        p = os.path.join(src, "pmc", f"{key}_{cn}.csv")
        if not os.path.exists(p):
            continue
        shutil.copy(p, dst)
        for r in csv.DictReader(open(p)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()     # templated kernels print as "void zj_x_kernel_t<false>(...)"
            if name.startswith("zj_"):
                rec[name][k].append(float(r["Counter_Value"]) * 1024.0)
    summ = {}
    stamp = None
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):                     # the driver's own line of the pass: which build, which route
        jp = os.path.join(src, "pmc", f"{key}_{cn}_driver.json")
        if os.path.exists(jp):
            for line in open(jp):
                if line.startswith("{"):
                    dj = json.loads(line); stamp = stamp or dj.get("build_stamp"); shutil.copy(jp, dst)
    for name, v in rec.items():
        # the driver runs 2 calls (1 warm-up + 1); kernels launched several times per call (lists A / B / S): the largest launch
        fetch = max(v["fetch"]) if v["fetch"] else 0.0
        write = max(v["write"]) if v["write"] else 0.0
        extra = {}
        if ("match_run" in name or "match_wide" in name) and "noflags" not in key:
            # the level-3 match kernels with need flags: counter collection serialises the kernels and now and then lets a call's match kernel go AHEAD of its flag kernel
            # (round 6: the metric key's WRITE_SIZE pass, second call — 120.8 GB, exactly the no-flags figure, against 67.6 GB in the call where the order held and in every
            # pass of key 2, the same command).  The figure quoted is the full-size launch with its flags in place: the smallest launch that is at least half the largest.
            pick = lambda xs: min(x for x in xs if x >= 0.5 * max(xs)) if xs else 0.0
            if v["fetch"] and pick(v["fetch"]) != fetch or v["write"] and pick(v["write"]) != write:
                extra = {"launches_fetch_bytes": v["fetch"], "launches_write_bytes": v["write"], "picked": "the full-size launch whose flag kernel ran ahead of it (the smallest launch >= half the largest)"}
            fetch, write = pick(v["fetch"]), pick(v["write"])
        summ[name] = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write, "build_stamp": stamp, **extra}
    out[key] = summ
    for extra in (f"{key}_kernel_stats.csv", f"{key}_driver.json"):     # the rocprofv3 --kernel-trace --stats summary of the same driver command
        ep = os.path.join(src, extra)
        if os.path.exists(ep):
            shutil.copy(ep, os.path.join(ROOT, "profiles", f"{rnd}_{extra}"))
with open(os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps({k: {kk: round(vv["hbm_bytes_per_launch"] / 1e9, 2) for kk, vv in v.items() if isinstance(vv, dict)} for k, v in out.items() if isinstance(v, dict) and k not in ("calibration",)}, indent=1))
