#!/usr/bin/env python3
"""Generates the legacy-program fixtures by RUNNING the reference's legacy program (oracle/_ref/old-diagnose,
built by oracle/Makefile from src/old-diagnose, unmodified; this container only):

  legacy_efficiency_64x48.json, legacy_64x48_fields.npz   (round 1: efficiency.txt + the [0_B0] fields of case A)
  legacy_64x48_tendency.npz       case A: CYLINDRICAL-TENDENCY-DENSITY_NORMAL-BAROTROPIC, unit domain, no boundary
                                  files, strategy 1 -- EVERY file the program writes (old-diagnose/diagnose.f90:389-724)
  legacy_64x48_baroall_bc.npz     case B: ...-BARO_ALL with psi and chi boundary files on a 2e5 m x 1.5e4 m domain,
                                  strategy 2 (relative criterion): all nine solves, both exchange-conversion methods,
                                  the complete efficiency.txt (:779-841)

Purpose: the legacy program is the only place the reference assembles the
source terms (A11), forecasts theta (Stage II) and computes the scalar energy-efficiency numbers, so its files pin the
oracle's restatements BITWISE and its efficiency.txt to the 8 digits it prints.  The legacy program has undefined
behaviour; these cases keep it inert:
  * Q_in / F_in are allocated (nr-1,nz-1) but read as nr x nz records: on this small grid the overflow lands in heap
    chunks that are rewritten later, and what is integrated is the SHEARED field
    Q_eff(i,j) = file_flat[(j-1)*(nr-1) + (i-1)], which the tests emulate;
  * m2 reads m2(0,j) and rhoC_C with stale indices (:361-366): F = 0 makes sqrt(m2)*F vanish whatever m2 holds;
  * sum_Qeta_0_0 is only assigned with a chi boundary file (:563): case A's efficiency.txt never prints it, case B has
    the file;
  * cal_exchange_conversion declares r, dr, dz INTEGER (:1146): on case B's metre-sized domain the truncation is
    deterministic (r = int((ra(i)+ra(i+1))/2), dr = int(ra(2)-ra(1)), dz = int(za(2)-za(1))) and is restated as such.
"""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orchelp as H  # noqa: E402

OUT = Path(__file__).resolve().parent
NR, NZ = 64, 48
RELAX = re.compile(r"Relaxation uses\s+(\d+)\s+steps\. Final residue is\s+(\S+)\s*\.")
SHAPES = {"O": (NZ, NR), "A": (NZ, NR - 1), "B": (NZ - 1, NR - 1), "C": (NZ - 1, NR), "sA": (NZ - 2, NR - 1), "sC": (NZ - 1, NR - 2)}


def q_file(nr=NR, nz=NZ):
    r = np.linspace(0, 1, nr)[None, :]
    z = np.linspace(0, 1, nz)[:, None]
    return np.exp(-((r - 0.3) / 0.1) ** 2 - ((z - 0.5) / 0.2) ** 2).astype(np.float32)   # nr x nz record


def case_A():
    A, B, Cc, _ = H.member_inputs(NR, NZ, 0)
    return dict(name="legacy_64x48_tendency", mode="CYLINDRICAL-TENDENCY-DENSITY_NORMAL-BAROTROPIC", testing_dt=100.0,
                domain=(0.0, 1.0, 0.0, 1.0), A=A, B=B, C=Cc, Q=q_file(), F=np.zeros((NZ, NR), np.float32),
                psi_solver=(1, 1e-7, 2000, 1.0), chi_solver=(1, 1e-7, 2000, 1.0), psi_bc=None, chi_bc=None)


def case_B():
    A, B, Cc, bc = H.rich_inputs(NR, NZ, seed=21)
    r = np.linspace(0, 1, NR)[None, :]
    z = np.linspace(0, 1, NZ)[:, None]
    chi_bc = (1e-3 * np.sin(np.pi * z) * (0.5 + r)).astype(np.float32)
    psi_bc = (2e-2 * np.cos(np.pi * z) * r * (1.0 - 0.5 * r)).astype(np.float32)
    return dict(name="legacy_64x48_baroall_bc", mode="CYLINDRICAL-TENDENCY-DENSITY_NORMAL-BARO_ALL", testing_dt=3.0e4,
                domain=(0.0, 2.0e5, 0.0, 1.5e4), A=A, B=B, C=Cc, Q=(2.0 * q_file()).astype(np.float32),
                F=np.zeros((NZ, NR), np.float32), psi_solver=(2, 2e-3, 5000, 1.0), chi_solver=(2, 2e-3, 5000, 0.9),
                psi_bc=psi_bc, chi_bc=chi_bc)


def run_legacy(c):
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "out").mkdir()
        for k in ("A", "B", "C", "Q", "F"):
            np.ascontiguousarray(c[k], np.float32).tofile(td / f"{k}.bin")
        lines = [c["mode"], "%.9e // testing_dt" % c["testing_dt"], " ".join("%.9e" % v for v in c["domain"]), "%d %d" % (NR, NZ),
                 ".", "out", "A.bin", "B.bin", "C.bin", "Q.bin", "F.bin",
                 "%d %.9e %d %.9e" % c["psi_solver"], "%d %.9e %d %.9e" % c["chi_solver"]]
        for key in ("psi_bc", "chi_bc"):
            if c[key] is None:
                lines.append("no")
            else:
                np.ascontiguousarray(c[key], np.float32).tofile(td / f"{key}.bin")
                lines += ["yes", f"{key}.bin"]
        (td / "diag.txt").write_text("\n".join(lines) + "\n")
        with open(td / "diag.txt") as fin:
            p = subprocess.run([str(H.REF_DIR / "old-diagnose")], stdin=fin, cwd=td, capture_output=True, text=True, timeout=600)
        eff = (td / "out" / "efficiency.txt").read_text()
        files = {}
        for fn in sorted((td / "out").glob("*.bin")):
            raw = np.fromfile(fn, np.float32)
            tag = fn.stem.rsplit("-", 1)[-1]
            shp = SHAPES.get(tag) if fn.stem != "J-B" else SHAPES["B"]
            if fn.stem.startswith("bndconv"):
                shp = (2, NR - 1)
            assert shp is not None and raw.size == shp[0] * shp[1], (fn.name, raw.size, shp)
            files[fn.name] = raw.reshape(shp)
    return p, eff, files


def main():
    for c in (case_A(), case_B()):
        p, eff, files = run_legacy(c)
        relax = [(int(n), r) for n, r in RELAX.findall(p.stdout)]
        print(c["name"], "rc", p.returncode, relax)
        print(eff)
        meta = {"mode": c["mode"], "testing_dt": c["testing_dt"], "domain": list(c["domain"]), "psi_solver": list(c["psi_solver"]),
                "chi_solver": list(c["chi_solver"]), "relax": relax, "efficiency_txt": eff, "returncode": p.returncode,
                "sum_dtheta_dt_line": [ln for ln in p.stdout.splitlines() if "Max dtheta_dt" in ln]}
        arrs = {"in_" + k: np.ascontiguousarray(c[k], np.float32) for k in ("A", "B", "C", "Q", "F")}
        for key in ("psi_bc", "chi_bc"):
            if c[key] is not None:
                arrs["in_" + key] = c[key]
        for fn, a in files.items():
            arrs["out_" + fn] = a
        np.savez_compressed(OUT / (c["name"] + ".npz"), meta=np.array(json.dumps(meta)), **arrs)
        if c["name"] == "legacy_64x48_tendency":        # the round-1 fixtures (same run)
            rec = {"nr": NR, "nz": NZ, "efficiency_txt": eff, "eta_0_B0_md5": H.md5(files["eta-[0_B0]-A.bin"]),
                   "rchi_0_B0_md5": H.md5(files["rchi-[0_B0]-O.bin"]), "returncode": p.returncode}
            np.savez_compressed(OUT / "legacy_64x48_fields.npz", eta_0_B0=files["eta-[0_B0]-A.bin"], rchi_0_B0=files["rchi-[0_B0]-O.bin"])
            (OUT / "legacy_efficiency_64x48.json").write_text(json.dumps(rec, indent=1) + "\n")


if __name__ == "__main__":
    main()
