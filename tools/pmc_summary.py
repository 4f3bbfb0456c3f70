#!/usr/bin/env python3
"""Condense the passes of tools/pmc_r5.sh (gpurun_out/<tag>/<group>_{pass1..4,kstats}.txt) into one JSON: per kernel the HBM-side bytes
per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KiB at the L2 <-> fabric boundary, MALL hits included -- MI355X_MICROARCH.md), the
matrix pipe's busy fraction (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD), VALU / LDS instructions per MFMA, LDS bank
conflict share and the wave-cycle split.   usage: python tools/pmc_summary.py gpurun_out/<tag> > profiles/r5/pmc_kernels.json"""
import json, os, re, sys

NSIMD, NXCD = 1024, 8


def parse(path):
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for l in open(path):
        if not l.startswith(" "):
            cur = l.strip()
            out.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=(\S+)", l)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(3))
                out[cur]["_n"] = int(m.group(2))
    return out


def main():
    root = sys.argv[1]
    res = {}
    for g in sorted({f.split("_pass")[0] for f in os.listdir(root) if "_pass" in f}):
        c = {}
        for i in (1, 2, 3, 4):
            for k, v in parse(os.path.join(root, f"{g}_pass{i}.txt")).items():
                c.setdefault(k, {}).update(v)
        dur = {}
        ks = os.path.join(root, f"{g}_kstats.txt")
        if os.path.exists(ks):
            for l in open(ks):
                m = re.match(r"(.*?)\s+calls\s+(\d+) avg\s+([\d.]+) us", l)
                if m:
                    dur[m.group(1).strip()] = float(m.group(3))
        for k, v in c.items():
            if not k.startswith(("k_", "_ZN5craft")) or "FETCH_SIZE" not in v:
                continue
            e = {"group": g, "launches_averaged": v.get("_n"), "kernel_trace_us": dur.get(k),
                 "fetch_kib_raw": v["FETCH_SIZE"], "write_kib_raw": v.get("WRITE_SIZE"), "fetch_correction": 2.0,
                 "hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v.get("WRITE_SIZE", 0.0) * 1024)}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
                e["mfma_busy"] = round((v["SQ_VALU_MFMA_BUSY_CYCLES"] / NSIMD) / (v["GRBM_GUI_ACTIVE"] / NXCD), 4)
                e["gpu_cycles_per_launch"] = v["GRBM_GUI_ACTIVE"] / NXCD
            if v.get("SQ_INSTS_MFMA"):
                e["mfma_insts"] = v["SQ_INSTS_MFMA"]
                e["valu_per_mfma"] = round(v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"], 3)
                e["lds_per_mfma"] = round(v.get("SQ_INSTS_LDS", 0.0) / v["SQ_INSTS_MFMA"], 3)
                e["salu_per_mfma"] = round(v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_MFMA"], 3)
            if v.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
            if v.get("SQ_WAVE_CYCLES"):
                w = v["SQ_WAVE_CYCLES"]
                e["wave_cycles_split"] = {"active_inst": round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / w, 3), "wait_inst": round(v.get("SQ_WAIT_INST_ANY", 0.0) / w, 3),
                                          "wait_any": round(v.get("SQ_WAIT_ANY", 0.0) / w, 3)}
            res.setdefault(k, []).append(e)
    # the shape tools/run_kernel.py ran at (its defaults unless the environment said otherwise)
    out = {"shape": {"B": int(os.environ.get("B", 4)), "H8": int(os.environ.get("H8", 56)), "W8": int(os.environ.get("W8", 128))},
           "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2 + GRBM, each its own run) + a --kernel-trace --stats run of "
                     "tools/run_kernel.py <group> mixed; tools/pmc_r5.sh", "kernels": res}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
