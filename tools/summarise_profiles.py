"""Condense the rocprofv3 output of tools/profile_round.sh (under gpurun_out/) into profiles/.

    python tools/summarise_profiles.py r01

writes profiles/<tag>_bench_stdout.json, <tag>_bench_kernel_stats.csv (rocprofv3 --stats, verbatim),
<tag>_bench_pmc_fetch_write.csv (per-kernel averages of the two PMC passes) and
<tag>_pmc_traffic.json (the prop kernel's HBM-side bytes per launch, read by bench.py)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True)
    if not hits:
        raise SystemExit(f"nothing matches gpurun_out/{pattern}")
    return sorted(hits)[-1]


def pmc_avgs(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        a = acc[row["Kernel_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    line = [l for l in open(one(f"{tag}_bench_stdout.json")) if l.startswith("{")][-1]
    bench = json.loads(line)
    json.dump(bench, open(os.path.join(prof, f"{tag}_bench_stdout.json"), "w"), indent=1)
    shutil.copy(one(f"{tag}_prof/**/*kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    try:
        shutil.copy(one(f"{tag}_prof/**/*domain_stats.csv"), os.path.join(prof, f"{tag}_bench_domain_stats.csv"))
    except SystemExit:
        pass
    fetch = pmc_avgs(one(f"{tag}_pmc_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
    write = pmc_avgs(one(f"{tag}_pmc_write/**/*counter_collection.csv"), "WRITE_SIZE")
    with open(os.path.join(prof, f"{tag}_bench_pmc_fetch_write.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "launches"])
        for k in sorted(set(fetch) | set(write)):
            w.writerow([k, fetch.get(k, (0, 0))[0], write.get(k, (0, 0))[0], max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])])
    # LDS bank conflicts: cycles stalled by conflicts / cycles the LDS was busy, per kernel
    for leg, name in (("pmc_lds", "bench"), ("pmc_lds_gp100", "gp100")):
        try:
            path = one(f"{tag}_{leg}/**/*counter_collection.csv")
        except SystemExit:
            continue
        conf, act, ins = (pmc_avgs(path, c) for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"))
        with open(os.path.join(prof, f"{tag}_{name}_pmc_lds.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "SQ_LDS_BANK_CONFLICT_avg", "SQ_LDS_IDX_ACTIVE_avg", "SQ_INSTS_LDS_avg",
                        "conflict_cycles_per_active_cycle", "launches"])
            for k in sorted(act):
                if "fbsmi" not in k:
                    continue
                a = act[k][0]
                w.writerow([k, conf.get(k, (0, 0))[0], a, ins.get(k, (0, 0))[0],
                            (conf.get(k, (0, 0))[0] / a) if a else 0.0, act[k][1]])
    try:
        shutil.copy(one(f"{tag}_prof_gp100/**/*kernel_stats.csv"), os.path.join(prof, f"{tag}_gp100_kernel_stats.csv"))
    except SystemExit:
        pass
    # instruction mix per wave of the step kernels
    valu_json = None
    try:
        path = one(f"{tag}_pmc_valu/**/*counter_collection.csv")
        names = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES")
        cols = {n: pmc_avgs(path, n) for n in names}
        valu_json = {}
        with open(os.path.join(prof, f"{tag}_bench_pmc_valu.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "waves_per_launch"] + [n + "_per_wave" for n in names[:-1]] + ["launches"])
            for k in sorted(cols["SQ_WAVES"]):
                if "fbsmi" not in k:
                    continue
                wv = cols["SQ_WAVES"][k][0]
                per = [cols[n].get(k, (0, 0))[0] / wv if wv else 0.0 for n in names[:-1]]
                w.writerow([k, wv] + per + [cols["SQ_WAVES"][k][1]])
                if "k_lg_prop" in k or "<1, 0>" in k or "<1, 0, " in k:   # the step kernels (MODE 0)
                    valu_json[k] = {"waves_per_launch": wv, "valu_per_wave": per[0], "salu_per_wave": per[1],
                                    "launches": cols["SQ_WAVES"][k][1]}
    except SystemExit:
        pass
    # FETCH_SIZE against a known byte count (tools/fetch_calib.hip reads 512 MiB per launch)
    calib = None
    try:
        cal = pmc_avgs(one(f"{tag}_fetch_calib/**/*counter_collection.csv"), "FETCH_SIZE")
        calib = {}
        for kname, (kb, cnt) in cal.items():
            width = "16_bytes_per_lane" if "dwordx4" in kname else "4_bytes_per_lane"
            calib[width] = {"kernel": kname, "FETCH_SIZE_KB": kb, "true_KB": 524288.0, "true_over_reported": 524288.0 / kb,
                            "launches": cnt}
        json.dump(calib, open(os.path.join(prof, f"{tag}_fetch_calibration.json"), "w"), indent=1)
    except SystemExit:
        pass
    try:
        shutil.copy(one(f"{tag}_prof_em/**/*kernel_stats.csv"), os.path.join(prof, f"{tag}_em_kernel_stats.csv"))
        emj = [l for l in open(one(f"{tag}_pmc_em.json")).read().split("\n{", 1)]
        txt = open(one(f"{tag}_pmc_em.json")).read()
        json.dump(json.loads(txt[txt.index("{"):]), open(os.path.join(prof, f"{tag}_em_pmc.json"), "w"), indent=1)
    except (SystemExit, ValueError):
        pass
    cfg = bench["config"]
    prop = [k for k in fetch if "k_lg_prop" in k]
    if not prop:
        raise SystemExit("no k_lg_prop kernel in the PMC pass")
    k = max(prop, key=lambda n: fetch[n][1])
    fkb, wkb = fetch[k][0], write.get(k, (0.0, 0))[0]
    f4 = calib["4_bytes_per_lane"]["true_over_reported"] if calib and "4_bytes_per_lane" in calib else None
    traffic = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                   "--steps 1 --warmup 1 --no-cpu-baseline --no-single-chain --batch-scan '' --image-steps 0 --sharded-steps 0 --sharded-lg-steps 0",
        "workload_text": cfg.get("workload"),
        "workload": {"nparticles": cfg.get("nparticles"), "nsteps": cfg.get("nsteps"), "nchains": cfg.get("nchains")},
        "chains_per_launch": bench["roofline"].get("chains_per_launch", cfg.get("nchains")),
        "kernel": k.replace("void fbsmi::", ""),
        "fetch_correction_4_bytes_per_lane": f4,
        "bytes_per_launch": ((fkb * (f4 or 1.0)) + wkb) * 1024.0,
        "k_lg_prop_FETCH_SIZE_KB": fkb,
        "k_lg_prop_WRITE_SIZE_KB": wkb,
        "k_lg_prop_bytes_per_launch": (fkb + wkb) * 1024.0,
        "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
        "note": "Counters in KB = 1024 B; bytes_per_launch = FETCH_SIZE x the factor measured by tools/fetch_calib.hip for "
                "4-byte-per-lane reads on this box (fetch_correction_4_bytes_per_lane; 1.0 if the calibration is missing) + "
                "WRITE_SIZE.  The counters sit on the memory side of the XCD L2s (Infinity-Cache hits included).",
    }
    if valu_json:
        pk = [kk for kk in valu_json if "k_lg_prop" in kk]
        if pk:
            top = max(pk, key=lambda kk: valu_json[kk]["launches"])
            traffic["k_lg_prop_valu_insts_per_launch"] = valu_json[top]["valu_per_wave"] * valu_json[top]["waves_per_launch"]
            traffic["valu_insts_per_launch"] = traffic["k_lg_prop_valu_insts_per_launch"]
            traffic["cycles_per_valu_inst"] = 2.7   # tools/valu_rate.hip / tools/int_rate.hip: 2.3 (two-operand) .. 4.2 (three-operand)
        traffic["step_kernels_instruction_mix"] = valu_json
    json.dump(traffic, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    print(open(os.path.join(prof, f"{tag}_bench_kernel_stats.csv")).read()[:1500])


if __name__ == "__main__":
    main()
