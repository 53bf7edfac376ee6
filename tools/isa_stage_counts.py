"""Static instruction counts per stage of the Euler step kernel, from the ISA: compiles fbs_amd/csrc/fbsmi_lg.hip with
-DFBSMI_STAMPS (the diagnostic build whose FBSMI_STAMP(i) markers read s_memrealtime between the stages), cuts
k_lg_prop1t<1, 4> at the markers and counts vector / scalar / LDS / global-memory instructions in every piece.
Both sides of every divergent branch are counted (a static count), so the total exceeds the executed 747 per wave that
rocprofv3's SQ_INSTS_VALU reports; the kernel's first four waves alone execute the tree-building piece.

    python tools/isa_stage_counts.py > profiles/r02_k_lg_prop1t_isa_stages.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN5fbsmi11k_lg_prop1tILi1ELi4EEEvNS_5LgDevEi"
STAGES = ["prologue: kernel arguments, chain view, key table",
          "entry loads (tile sums, tree nodes, reference row, tables) + the slot's normal (Threefry + erf_inv)",
          "tree levels above the tiles, J_prob[i*], search for the rotation J (first four waves; the others wait at its barriers)",
          "rotation source: w[src], u[src], the kill-test and redraw uniforms (two Threefry calls), LDS levels of the Cat(w) search",
          "killed slots: tree nodes of the tile (one round trip), last four leaves + candidate rows (another), ancestor; "
          "gather, affine drift, Euler-Maruyama, pin, log-weight",
          "stores of the new particle and log-weight",
          "tile (max, sumexp) of the new log-weights (exp evaluated in float64), publication"]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c",
               "-DFBSMI_STAMPS", "-Wno-unused-value", "-Wno-pass-failed", "-save-temps=obj", "-o", os.path.join(tmp, "lg.o"),
               os.path.join(ROOT, "fbs_amd", "csrc", "fbsmi_lg.hip")]
        subprocess.check_call(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "fbsmi_lg-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    start = next(i for i, l in enumerate(asm) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    body = asm[start:end + 1]
    cuts = [i for i, l in enumerate(body) if "s_memrealtime" in l]
    pieces = [(0, cuts[0])] + [(cuts[k], cuts[k + 1]) for k in range(len(cuts) - 1)] + [(cuts[-1], len(body))]
    out = []
    for k, (a, b) in enumerate(pieces):
        ins = [l.strip().split()[0] for l in body[a:b] if re.match(r"\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
        c = lambda pred: sum(1 for x in ins if pred(x))
        out.append({"stage": STAGES[k] if k < len(STAGES) else f"piece {k}", "isa_lines": [a, b],
                    "valu": c(lambda x: x.startswith("v_")), "valu_f64": c(lambda x: x.startswith("v_") and "f64" in x),
                    "valu_three_operand": c(lambda x: x.startswith(("v_fma", "v_alignbit", "v_xad", "v_add3", "v_lshl_add", "v_div_",
                                                                     "v_cndmask_b32_e64", "v_perm", "v_bfe", "v_mad"))),
                    "salu": c(lambda x: x.startswith("s_") and not x.startswith(("s_waitcnt", "s_barrier", "s_nop"))),
                    "lds": c(lambda x: x.startswith("ds_")), "global_loads": c(lambda x: x.startswith("global_load")),
                    "global_stores": c(lambda x: x.startswith("global_store")), "barriers": c(lambda x: x == "s_barrier"),
                    "waits": c(lambda x: x == "s_waitcnt")})
    json.dump({"kernel": "fbsmi::k_lg_prop1t<1, 4> (diagnostic build -DFBSMI_STAMPS; static counts, both sides of divergent branches)",
               "markers_found": len(cuts), "stages": out,
               "executed_per_wave_rocprofv3": "profiles/r02_bench_pmc_valu.csv: 747 VALU, 291 SALU per wave on average"},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
