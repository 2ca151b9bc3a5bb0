"""Config / result tooling around the host program.

  write_diag_config(path, setting)  -- diag.txt writer taking the same `setting` dict as the
        reference's xtt-lib-python/writeDiagConfig.py:24-50 (field names: :4-19), but writing
        numbers with %.9e so small tolerances survive (the reference's %f turns 1e-8 into
        0.000000, which silently disables the criterion, quirk 12) and with a correct SPHERICAL
        branch (the reference's has 4 %f for 3 values and raises).  Optional key 'Q_FILE' emits
        the extra heating-file line understood by xee-diagnose.
  read_efficiency(path)             -- efficiency.txt reader returning the same keys as
        xtt-lib-python/XEffReader.py:3-32 (semi_internal, internal, semi_cb1, wtheta,
        local_response, semi_total), keyed on the same line prefixes.
  empty_setting()                   -- the defaults of writeDiagConfig.emptySetting (:3-19).
  ensemble_member(nr, nz, m)        -- SURVEY.md 8d synthetic member m.

Pure Python/numpy; independent of the reference's Python files (which are not importable on the GPU box).
"""
import os

import numpy as np

TYPE = ("DYNAMIC_EFFICIENCY", "SECONDARY_CIRCULATION", "NONE")
GEOMETRY = ("CYLINDRICAL", "SPHERICAL")
DENSITY = ("DENSITY_NORMAL", "DENSITY_BOUSSINESQ")
OPERATOR_COMPLEXITY = ("BARO_ALL", "BAROCLINIC", "BAROTROPIC")


def empty_setting():
    return {
        "TYPE": "DYNAMIC_EFFICIENCY", "GEOMETRY": "CYLINDRICAL", "DENSITY": "DENSITY_NORMAL",
        "OPERATOR_COMPLEXITY": "BAROTROPIC",
        "POINTS": {"horizontal": 100, "vertical": 100},
        "DOMAIN_RANGE": {"horizontal": [0.0, 100.0], "vertical": [0.0, 100.0], "planet_radius": 6371000.0},
        "INPUT_FOLDER": ".", "OUTPUT_FOLDER": ".", "A_FILE": "A.bin", "B_FILE": "B.bin", "C_FILE": "C.bin",
        "FORCING_FILE": "forcing.bin", "BC_INIT_FILE": "bc_init.bin",
        "RESIDUE_CRITERIA": {"absolute": 1e-5, "relative": 1e-3, "alpha": 1.0, "max_iteration": 100000},
    }


def write_diag_config(path, setting):
    """Line order = src/diagnose/read-input.f90:2-89."""
    s = setting
    for key, allowed in (("TYPE", TYPE), ("GEOMETRY", GEOMETRY), ("DENSITY", DENSITY),
                         ("OPERATOR_COMPLEXITY", OPERATOR_COMPLEXITY)):
        if s[key] not in allowed:
            raise ValueError(f"{key}={s[key]!r} not in {allowed}")
    dom = s["DOMAIN_RANGE"]
    lines = ["%s-%s-%s-%s   // type-geometry-density-operator_complexity"
             % (s["TYPE"], s["GEOMETRY"], s["DENSITY"], s["OPERATOR_COMPLEXITY"])]
    if s["GEOMETRY"] == "CYLINDRICAL":
        lines.append("%.9e %.9e %.9e %.9e // domain size" % (dom["horizontal"][0], dom["horizontal"][1],
                                                              dom["vertical"][0], dom["vertical"][1]))
    else:
        lines.append("%.9e %.9e %.9e // planet radius, vertical range" % (dom["planet_radius"], dom["vertical"][0],
                                                                          dom["vertical"][1]))
    lines.append("%d %d // grid points" % (s["POINTS"]["horizontal"], s["POINTS"]["vertical"]))
    lines.append("%s    // input folder" % s["INPUT_FOLDER"])
    lines.append("%s    // output folder" % s["OUTPUT_FOLDER"])
    lines.append("%s    // file: A" % s["A_FILE"])
    lines.append("%s    // file: B" % s["B_FILE"])
    lines.append("%s    // file: C" % s["C_FILE"])
    if s["TYPE"] == "SECONDARY_CIRCULATION":
        lines.append("%s    // file: forcing" % s["FORCING_FILE"])
    lines.append("%s    // file: boundary condition and initial guess" % s["BC_INIT_FILE"])
    rc = s["RESIDUE_CRITERIA"]
    lines.append("%.9e %.9e %d %.9e // residue absolute, residue relative, max iteration, alpha"
                 % (rc["absolute"], rc["relative"], rc["max_iteration"], rc["alpha"]))
    if s.get("Q_FILE"):
        lines.append("%s    // file: heating Q on B cells (xee-diagnose extension -> efficiency.txt)" % s["Q_FILE"])
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


_PREFIXES = (
    (" eta [L(B=0)    = 0]      w/  boundary", ("semi_internal",)),
    (" eta [L(B=0)    = dB]     wo/ boundary", ("semi_internal", "internal")),
    (" eta [L(B=0)    = B0]     wo/ boundary", ("semi_internal", "internal")),
    (" bndconv [L(B=0) = B0dB]   w/ boundary", ("semi_cb1",)),
    (" wtheta [L(B=0)    = J F] w/  boundary", ("wtheta",)),
)


def read_efficiency(path):
    """Same keys and the same line-prefix matching as XEffReader.readEfficiency."""
    if not os.path.exists(path):
        raise IOError(path)
    eff = {k: 0.0 for k in ("semi_internal", "semi_cb1", "internal", "wtheta", "local_response")}
    with open(path) as fh:
        for line in fh:
            if line.startswith(" Local heat response (sum Q / sum dtheta_dt)"):
                eff["local_response"] += float(line.split(":")[1])
                continue
            for prefix, keys in _PREFIXES:
                if line.startswith(prefix):
                    val = float(line.split(":")[1].split(",")[1])
                    for k in keys:
                        eff[k] += val
                    break
    eff["semi_total"] = eff["semi_internal"] + eff["semi_cb1"]
    return eff


# the sweep's cases under the legacy labels (old-diagnose/diagnose.f90:779-841); the three XEffReader keys on
# are written exactly as it expects them, the baroclinic-operator and thermal lines are additional
SWEEP_LABELS = {
    ("BAROTROPIC", "rim_bc"): " eta [L(B=0)    = 0]      w/  boundary",
    ("BAROTROPIC", "minus_B"): " eta [L(B=0)    = B0]     wo/ boundary",
    ("BAROTROPIC", "thermal"): " eta [L(B=0)    = J Q]    wo/ boundary",
    ("BAROCLINIC", "rim_bc"): " eta [L(B=B0dB) = 0]      w/  boundary",
    ("BAROCLINIC", "minus_B"): " eta [L(B=B0dB) = B0]     wo/ boundary",
    ("BAROCLINIC", "thermal"): " eta [L(B=B0dB) = J Q]    wo/ boundary",
}


def write_efficiency(path, sum_Q, entries, elapsed=None):
    """efficiency.txt in the legacy layout: `entries` = [(label, sum_Qeta)], each written as
    ` <label> : <sum_Qeta> , <sum_Qeta / sum_Q>` with ES16.9 numbers (an fp32 value round-trips)."""
    with open(path, "w") as fh:
        if elapsed is not None:
            fh.write(" Time elapsed (sec)                          : %16.9E\n" % elapsed)
        fh.write(" sum Q                                       : %16.9E\n" % float(sum_Q))
        for label, v in entries:
            ratio = float(np.float32(v) / np.float32(sum_Q)) if sum_Q != 0 else float("nan")
            fh.write("%s : %16.9E , %16.9E\n" % (label, float(v), ratio))


def write_sweep_efficiency(records, outdir, op_names=("BAROTROPIC", "BAROCLINIC"), rhs_names=("minus_B", "thermal", "rim_bc")):
    """One efficiency.txt per timestep of a source-term sweep (sweep.run records: case id = 6 t + 3 op + rhs)
    under `outdir/t%04d/`; returns the paths."""
    per = len(op_names) * len(rhs_names)
    paths = []
    for t in range(len(records) // per):
        block = {int(r["case_id"]) - per * t: r for r in records[per * t: per * (t + 1)]}
        d = os.path.join(outdir, "t%04d" % t)
        os.makedirs(d, exist_ok=True)
        entries = []
        for o, op in enumerate(op_names):
            for k, rhs in enumerate(rhs_names):
                entries.append((SWEEP_LABELS[(op, rhs)], block[len(rhs_names) * o + k]["sum_Qeta"]))
        path = os.path.join(d, "efficiency.txt")
        write_efficiency(path, block[0]["sum_Q"], entries)
        paths.append(path)
    return paths


def read_efficiency_raw(path):
    """All `label : value[, ratio]` lines -> {label: (value, ratio or None)} (labels stripped)."""
    out = {}
    with open(path) as fh:
        for line in fh:
            if ":" not in line or line.lstrip().startswith("#"):
                continue
            label, rest = line.split(":", 1)
            parts = [p.strip() for p in rest.split(",")]
            out[label.strip()] = (float(parts[0]), float(parts[1]) if len(parts) > 1 else None)
    return out


def ensemble_member(nr, nz, m):
    """A, B, C, bc_init of synthetic member m (float64 math then float32), SURVEY.md 8d."""
    r = np.linspace(0.0, 1.0, nr)
    z = np.linspace(0.0, 1.0, nz)
    B = 1e-2 * np.sin(2.0 * np.pi * (1 + m % 4) * r)[None, :] * np.sin(np.pi * (1 + m % 5) * z)[:, None]
    if m > 0:
        B = B + 1e-4 * np.random.default_rng(20140911 + m).uniform(-1.0, 1.0, size=(nz, nr))
    one = np.ones((nz, nr), np.float32)
    return one, B.astype(np.float32), one.copy(), np.zeros((nz, nr), np.float32)
