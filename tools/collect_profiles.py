#!/usr/bin/env python3
"""Copy what is to be judged from gpurun_out/<tag>/ (tools/profile_round.sh + the bench lines + the probes) into
profiles/<round>/: kernel stats, trimmed PMC counter rows (sweep kernels and the two FETCH calibration kernels only),
bench lines, probe outputs; then pmc_summary.json and ../traffic.json (tools/pmc_summarize.py).
usage: python tools/collect_profiles.py gpurun_out/<tag> profiles/<round>"""
import csv
import glob
import shutil
import subprocess
import sys
from pathlib import Path

KEEP = ("k_sweep", "k_negate", "k_cal_eta")


def main():
    src, dst = Path(sys.argv[1]), Path(sys.argv[2])
    (dst / "pmc").mkdir(parents=True, exist_ok=True)
    newest = lambda pattern: sorted(glob.glob(pattern, recursive=True), key=lambda f: Path(f).stat().st_mtime)[-1:]
    for f in newest(str(src / "stats" / "**" / "*kernel_stats.csv")):
        shutil.copy(f, dst / "final_kernel_stats.csv")
    for f in newest(str(src / "stats_baroclinic" / "**" / "*kernel_stats.csv")):
        shutil.copy(f, dst / "final_kernel_stats_baroclinic.csv")
    for wl_dir in sorted((src / "pmc").glob("*")):
        for p in ("fetch", "write", "sq1", "sq2", "tcc"):
            files = newest(str(wl_dir / p / "**" / "*counter_collection.csv"))
            if not files:
                continue
            rows = list(csv.DictReader(open(files[0])))
            kept = [r for r in rows if any(k in r["Kernel_Name"] for k in KEEP)]
            with open(dst / "pmc" / f"{wl_dir.name}_{p}_counter_collection.csv", "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(kept)
    for f in list(src.glob("bench_*.json")) + list(src.glob("probe_*.txt")) + list(src.glob("timeline_*.txt")):
        if f.name != "bench_under_rocprof.json":
            shutil.copy(f, dst / f.name)
    subprocess.check_call([sys.executable, str(Path(__file__).parent / "pmc_summarize.py"), str(src), str(dst)], stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    main()
