#!/usr/bin/env python3
"""Turns the rocprofv3 counter CSVs of tools/profile_round.sh into profiles/<round>/pmc_summary.json and
profiles/traffic.json (the per-launch HBM bytes bench.py quotes as roofline.traffic).

    python tools/pmc_summarize.py gpurun_out/<tag> profiles/<round>

Layout read:  <tag>/pmc/<workload>/<pass>/**/*counter_collection.csv   pass in {fetch, write, sq1, sq2}
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 PMC sections): FETCH_SIZE and WRITE_SIZE in separate
passes, KiB per dispatch.  On gfx950 FETCH_SIZE tallies every fabric read request at 64 B although full lines are
128-B requests (the guide: "reports exactly 1/2 of the bytes of a wide coalesced streaming read -- double it"), so the
true read bytes lie between the tally and twice the tally; how far depends on the kernel's mix of partial and
full-line requests.  Two kernels of known byte count are measured in the same run as a check of the rule (k_negate,
16 B/lane: the exact 1/2; k_cal_eta, 4 B/lane with unaligned neighbours: 0.73), and the sweep's read bytes are taken as
    min(2 x tally, what its tiles can request at most)   but not below   what a pass over the planes must read at least
(both physical bounds are computed from the grid and the tiling).  With the XCD-aware tile order of round 2 the two
ends meet: 2 x tally = the compulsory bytes to 0.2 %, i.e. every line is fetched once, as a full line.
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

SHAPES = {"ensemble": (1024, 512, 32, "BAROTROPIC"), "baroclinic": (1024, 512, 32, "BAROCLINIC"),
          "single": (1024, 512, 1, "BAROTROPIC"), "highres": (4096, 2048, 1, "BAROTROPIC")}
# bytes per lane of the dominant kernel's global loads (decides which calibration applies)
LOAD_WIDTH = {"k_sweep_z": 16, "k_sweep_q": 4, "k_sweep_tb4": 4, "k_sweep_tb": 4, "k_sweep_stream": 16, "k_sweep_p": 8}


def read_pass(d):
    """-> {kernel_name: {counter: [values per dispatch]}}, {kernel_name: meta}"""
    vals = defaultdict(lambda: defaultdict(list))
    meta = {}
    files = sorted(glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
    rows_of = defaultdict(list)
    for fn in files[-1:]:                                  # (the newest run only, should the directory hold several)
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                rows_of[row["Kernel_Name"]].append(row)
    for k, rows in rows_of.items():
        # a kernel's dispatches of its most frequent grid only: the sweep's short launch at the head of a block has fewer,
        # larger tiles and would blur the per-launch means
        grids = [int(r["Grid_Size"]) for r in rows]
        mode = max(set(grids), key=grids.count)
        for row in rows:
            if int(row["Grid_Size"]) != mode:
                continue
            vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = {"grid": int(row["Grid_Size"]), "wg": int(row["Workgroup_Size"]), "vgpr": int(row["VGPR_Count"]),
                       "agpr": int(row["Accum_VGPR_Count"]), "lds": int(row["LDS_Block_Size"]), "scratch": int(row["Scratch_Size"])}
    return vals, meta


def short(name):
    for k in ("k_sweep_z", "k_sweep_q", "k_sweep_p", "k_sweep_tb4", "k_sweep_tb", "k_sweep_stream"):
        if k + "<" in name:
            return k
    return None


def main():
    src, dst = Path(sys.argv[1]), Path(sys.argv[2])
    dst.mkdir(parents=True, exist_ok=True)
    summary, entries = {}, []
    for wl, (nr, nz, M, op) in SHAPES.items():
        base = src / "pmc" / wl
        if not base.exists():
            continue
        per = {}
        for p in ("fetch", "write", "sq1", "sq2", "tcc"):
            if (base / p).exists():
                per[p] = read_pass(base / p)
        if "fetch" not in per:
            continue
        fv, fmeta = per["fetch"]
        # dominant sweep kernel = the sweep instantiation with the most dispatches
        sweeps = [(len(v["FETCH_SIZE"]), k) for k, v in fv.items() if short(k)]
        if not sweeps:
            continue
        _, kname = max(sweeps)
        mean = lambda xs: sum(xs) / len(xs)
        plane_kib = ((nr + 63) // 64 * 64) * nz * 4 / 1024.0 * M
        cal = {}
        for k, v in fv.items():
            if "k_negate" in k:
                cal["x16"] = mean(v["FETCH_SIZE"]) / plane_kib
            if "k_cal_eta" in k:
                cal["x4"] = mean(v["FETCH_SIZE"]) / (nr * nz * 4 / 1024.0 * M)
        width = LOAD_WIDTH.get(short(kname), 4)
        corr = cal.get("x16", 0.5) if width == 16 else cal.get("x4", 0.735) if width == 4 else None
        if corr is None:            # 8 B/lane: not separately calibrated, bracket between the two
            corr = 0.5 * (cal.get("x16", 0.5) + cal.get("x4", 0.735))
        fetch = mean(fv[kname]["FETCH_SIZE"])
        wgs = fmeta[kname]["grid"] // fmeta[kname]["wg"]
        planes = 4 if op == "BAROTROPIC" else 6
        # generating planes that all members share are read from ONE copy (xee_timing_t.shared_planes, printed by the workload)
        shared = 0
        try:
            import re as _re
            m_ = _re.search(r"shared_planes=(\d+)", (base / "fetch.log").read_text())
            shared = bin(int(m_.group(1)) & (3 if op == "BAROTROPIC" else 15)).count("1") if m_ else 0
        except OSError:
            pass
        if short(kname) == "k_sweep_z":
            # the streaming sweep: one wave per (strip of 52 own columns, pair of chunks, member); every wave reads 64 columns of
            # its chunks' rows plus 2 T rows of run-in per chunk (and up to two fetch groups more), four planes
            T, W = (6, 52) if op == "BAROTROPIC" else (4, 56)   # (the nine-weight operator: four iterations per launch)
            if op != "BAROTROPIC":
                planes = 5                                        # psi, f, AP, CQ and b (it makes BX and BY itself)
            strips = -(-(nr - 2) // W)
            ML, P = M, 1
            for ml in (M, max(1, M // 2)):
                for pp in range(1, 65):
                    if -(-(strips * pp * ml) // 8) * 8 == wgs:
                        ML, P = ml, pp
            Lc = -(-(nz - 2) // (2 * P))
            lower = (planes - shared) * nr * nz * 4.0 * ML + shared * nr * nz * 4.0
            upper = strips * P * ML * 2 * (Lc + 2 * T + 6) * 64 * 4.0 * planes
        else:
            # physical bounds of the sweep's read bytes per launch (T = 8, 64 x H tiles stepping by 64 - 2T, H - 2T)
            T, H = 8, (64 if op == "BAROTROPIC" else 48)
            ntx, nty = -(-(nr - 1) // (64 - 2 * T)), -(-(nz - 1) // (H - 2 * T))
            # members per LAUNCH: a large batch is solved as two halves side by side (xee_timing_t.sweep_streams), each launch
            # then covers half the members -- read it off the grid (workgroups = tiles rounded up to a multiple of eight)
            ML = max(1, min(M, round(wgs / (ntx * nty))))
            lower = (planes - shared) * nr * nz * 4.0 * ML + shared * nr * nz * 4.0   # every plane byte once (a shared plane: one copy)
            upper = ntx * nty * ML * (planes * 64 * H + (1 if planes == 4 else 2) * 64) * 4.0  # every tile its whole footprint
        read_bytes = max(min(2.0 * fetch * 1024.0, upper), lower)
        corr = fetch * 1024.0 / read_bytes
        rec = {"workload": wl, "kernel": kname, "short": short(kname), "nr": nr, "nz": nz, "members": ML, "members_in_batch": M, "operator": op,
               "generating_planes_shared": shared,
               "dispatches": len(fv[kname]["FETCH_SIZE"]), **fmeta[kname],
               "FETCH_SIZE_kib": fetch, "fetch_calibration": cal, "fetch_correction": corr, "load_width_B": width,
               "read_bytes": read_bytes, "read_bytes_bounds": {"tally": fetch * 1024.0, "twice_tally": 2048.0 * fetch,
                                                                "compulsory": lower, "all_tile_footprints": upper}}
        if "write" in per and kname in per["write"][0]:
            rec["WRITE_SIZE_kib"] = mean(per["write"][0][kname]["WRITE_SIZE"])
            rec["bytes_per_launch"] = read_bytes + rec["WRITE_SIZE_kib"] * 1024.0
        for p in ("sq1", "sq2", "tcc"):
            if p in per and kname in per[p][0]:
                for c, xs in per[p][0][kname].items():
                    rec[c] = mean(xs)
        summary[wl] = rec
        if "bytes_per_launch" in rec:
            entries.append({k: rec[k] for k in ("workload", "short", "kernel", "nr", "nz", "members", "operator", "FETCH_SIZE_kib",
                                                "WRITE_SIZE_kib", "fetch_correction", "bytes_per_launch")}
                           | {c: rec[c] for c in rec if c.startswith("SQ_") or c.startswith("TCC_")})
    # provenance: the hash of the kernel sources the passes ran on (written on the GPU box by profile_round.sh /
    # pmc_only.sh) and the git HEAD at summary time -- bench.py quotes an entry only if its own sources hash the same
    bh = (src / "build_hash.txt").read_text().strip() if (src / "build_hash.txt").exists() else None
    try:
        import subprocess
        head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=Path(__file__).parent).stdout.strip() or None
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "xlab-ee-fortran_amd/csrc", "xlab-ee-fortran_amd/Makefile"],
                                    capture_output=True, text=True, cwd=Path(__file__).parent.parent).stdout.strip())
    except OSError:
        head, dirty = None, None
    for e_ in entries:
        e_["build_hash"] = bh
        e_["head"] = head
        e_["head_dirty"] = dirty
    (dst / "pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
    if entries:
        tj = dst.parent / "traffic.json"
        (tj).write_text(json.dumps({"_comment": "per-launch HBM bytes and SQ counters of the dominant sweep kernel, from the rocprofv3 PMC passes "
                                    f"summarised in {dst.name}/pmc_summary.json (tools/pmc_summarize.py); bench.py quotes the entry of its workload",
                                    "round": dst.name, "build_hash": bh, "head": head, "entries": entries}, indent=1) + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
