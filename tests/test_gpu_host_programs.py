"""GPU tests of the Fortran host side, byte for byte against the reference's golden output files:
  * bin/xee-diagnose            -- the Fortran 2008 host program (stdin config, .bin files in/out)
  * oracle/_ref/diagnose_dropin -- the REFERENCE program itself (main.f90 + its support modules)
    with only `module elliptic_tools` swapped for fortran/elliptic_tools.f90 -> libxee: the
    source-level drop-in proof (built in this container; the binary travels to the GPU box)."""
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import orchelp as H
import xee_pkg

pytestmark = pytest.mark.gpu

xee = xee_pkg.load()
from xlab_ee_fortran_amd import xee_tools as T  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "xlab-ee-fortran_amd" / "bin" / "xee-diagnose"
DROPIN = ROOT / "oracle" / "_ref" / "diagnose_dropin"
RELAX = re.compile(r"Relaxation uses\s+(\d+)\s+steps\. Final residue is\s+(\S+)\s*,\s*(\S+)")
CASES = ["p128x64_test1", "s64x48_baroall", "s64x48_seccirc", "s64x48_converge", "s64x48_spherical"]


def run_program(exe, workdir, g, cfg, q=None):
    nz, nr = g["in_A"].shape
    for name, key in (("A", "in_A"), ("B", "in_B"), ("C", "in_C"), ("bc_init", "in_bc")):
        g[key].tofile(workdir / f"{name}.bin")
    mode = tuple(cfg.get("mode", ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC")))
    s = T.empty_setting()
    s["TYPE"], s["GEOMETRY"], s["DENSITY"], s["OPERATOR_COMPLEXITY"] = mode
    dom = cfg.get("domain", (0.0, 1.0, 0.0, 1.0))
    if mode[1] == "CYLINDRICAL":
        s["DOMAIN_RANGE"] = {"horizontal": [dom[0], dom[1]], "vertical": [dom[2], dom[3]], "planet_radius": 0.0}
    else:
        s["DOMAIN_RANGE"] = {"horizontal": [0.0, 0.0], "vertical": [dom[1], dom[2]], "planet_radius": dom[0]}
    s["POINTS"] = {"horizontal": nr, "vertical": nz}
    s["RESIDUE_CRITERIA"] = {"absolute": cfg["r1"], "relative": cfg["r2"], "alpha": cfg["alpha"], "max_iteration": cfg["max_iter"]}
    if "in_forcing" in g.files:
        g["in_forcing"].tofile(workdir / "forcing.bin")
    if q is not None:
        q.tofile(workdir / "Q.bin")
        s["Q_FILE"] = "Q.bin"
    T.write_diag_config(str(workdir / "diag.txt"), s)
    with open(workdir / "diag.txt") as fin:
        p = subprocess.run([str(exe)], stdin=fin, cwd=workdir, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def check_outputs(workdir, g, rec, stdout):
    relax = [(int(a), b, c) for a, b, c in RELAX.findall(stdout)]
    assert [r[0] for r in relax] == [r[0] for r in rec["relax"]]
    for (_, r1, r2), (_, g1, g2) in zip(relax, rec["relax"]):
        assert float(r1) == pytest.approx(float(g1), rel=2e-7) and float(r2) == pytest.approx(float(g2), rel=2e-7)
    outs = [k[4:] for k in g.files if k.startswith("out_")]
    assert outs
    for fn in outs:
        got = np.fromfile(workdir / fn, np.float32)
        assert got.tobytes() == g["out_" + fn].tobytes(), fn
    assert (workdir / "result.txt").read_text().lstrip().startswith("Time elapsed (sec) :")


@pytest.mark.parametrize("name", CASES)
def test_fortran_host_program_matches_reference_files(name, golden_hashes, tmp_path):
    assert HOST.exists(), "build it: make -C xlab-ee-fortran_amd fortran"
    g = np.load(H.GOLDEN / f"{name}.npz")
    rec = golden_hashes[name]
    stdout = run_program(HOST, tmp_path, g, rec["cfg"])
    check_outputs(tmp_path, g, rec, stdout)


@pytest.mark.parametrize("name", ["p128x64_test1", "s64x48_baroall", "s64x48_seccirc"])
def test_reference_program_with_dropin_module(name, golden_hashes, tmp_path):
    if not DROPIN.exists():
        pytest.skip("oracle/_ref/diagnose_dropin not built (needs the reference sources)")
    g = np.load(H.GOLDEN / f"{name}.npz")
    rec = golden_hashes[name]
    stdout = run_program(DROPIN, tmp_path, g, rec["cfg"])
    check_outputs(tmp_path, g, rec, stdout)
    assert "Elliptic Tools:" in stdout


def test_efficiency_txt_extension(tmp_path):
    """Optional Q line -> efficiency.txt readable by the XEffReader-compatible reader; values are
    the oracle's order-exact fp32 sums (ES16.9 round-trips an fp32)."""
    g = np.load(H.GOLDEN / "s64x48_baroall.npz")
    nz, nr = g["in_A"].shape
    r = np.linspace(0, 1, nr - 1)[None, :]
    z = np.linspace(0, 1, nz - 1)[:, None]
    Q = np.exp(-((r - 0.3) / 0.1) ** 2 - ((z - 0.5) / 0.2) ** 2).astype(np.float32)
    cfg = dict(mode=("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BARO_ALL"),
               domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, max_iter=300, alpha=0.9)
    run_program(HOST, tmp_path, g, cfg, q=Q)
    raw = T.read_efficiency_raw(str(tmp_path / "efficiency.txt"))
    want = H.diagnose_case(g["in_A"], g["in_B"], g["in_C"], g["in_bc"], **cfg)
    prof = want["profiles"]
    sq = H.integrate_weight_B(Q, prof)
    assert np.float32(raw["sum Q"][0]).tobytes() == sq.tobytes()
    for v, label in (("BAROTROPIC", "eta [L(B=0)    = B0]     wo/ boundary"), ("BAROCLINIC", "eta [L(B=B0dB) = B0]     wo/ boundary")):
        sqe = H.cal_sum_Qeta(Q, want[v]["eta"], prof)
        assert np.float32(raw[label][0]).tobytes() == sqe.tobytes()
        assert np.float32(raw[label][1]).tobytes() == np.float32(sqe / sq).tobytes()
    eff = T.read_efficiency(str(tmp_path / "efficiency.txt"))
    assert eff["internal"] == pytest.approx(float(np.float32(H.cal_sum_Qeta(Q, want["BAROTROPIC"]["eta"], prof) / sq)), rel=1e-6)


# ---- ensemble / case-list mode of the Fortran host (bin/xee-ensemble) ---------------------------------------------

ENSEMBLE = ROOT / "xlab-ee-fortran_amd" / "bin" / "xee-ensemble"


@pytest.mark.parametrize("force_gather", [False, True])
def test_fortran_ensemble_host_equals_single_case_runs(force_gather, tmp_path):
    """N cases in ONE device ensemble (members stop on their own criterion) write, per case, byte for byte the files
    xee-diagnose writes when run on that case alone; the gathered records land in ensemble_summary.txt.  With
    XEE_ENSEMBLE_FORCE_GATHER the one-rank run goes through xee_comm_* / xee_gather_scalars (RCCL) for real."""
    assert ENSEMBLE.exists() and HOST.exists(), "build them: make -C xlab-ee-fortran_amd fortran"
    nr, nz, N = 96, 64, 5
    mode = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BARO_ALL")
    cfg = dict(mode=mode, domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1.0, r2=5e-3, alpha=1.0, max_iter=20000)
    rb = np.linspace(0, 1, nr - 1)[None, :]
    zb = np.linspace(0, 1, nz - 1)[:, None]
    Q = np.exp(-((rb - 0.3) / 0.1) ** 2 - ((zb - 0.5) / 0.2) ** 2).astype(np.float32)
    dirs, singles = [], []
    for k in range(N):
        A, B, Cc, bc = H.rich_inputs(nr, nz, seed=60 + k)
        g = {"in_A": A, "in_B": B, "in_C": Cc, "in_bc": bc}
        g = type("G", (dict,), {"files": list(g)})(g)
        d = tmp_path / f"case{k}"; d.mkdir()
        s = tmp_path / f"single{k}"; s.mkdir()
        run_program(HOST, s, g, cfg, q=Q)                            # the single-case reference run (also leaves the inputs)
        for name in ("A", "B", "C", "bc_init", "Q"):
            (d / f"{name}.bin").write_bytes((s / f"{name}.bin").read_bytes())
        dirs.append(d); singles.append(s)
    lines = ["-".join(mode), "%.9e %.9e %.9e %.9e" % cfg["domain"], f"{nr} {nz}",
             "%.9e %.9e %d %.9e // criteria" % (cfg["r1"], cfg["r2"], cfg["max_iter"], cfg["alpha"]), str(N)] + [str(d) for d in dirs]
    (tmp_path / "ensemble.txt").write_text("\n".join(lines) + "\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "XEE_RANK", "XEE_WORLD")}
    if force_gather:
        env["XEE_ENSEMBLE_FORCE_GATHER"] = "1"
    with open(tmp_path / "ensemble.txt") as fin:
        p = subprocess.run([str(ENSEMBLE)], stdin=fin, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    iters = set()
    for k in range(N):
        for fn in ("solver_a-sA.bin", "solver_b-B.bin", "solver_c-sC.bin", "eta-[BAROTROPIC]-A.bin", "rchi-[BAROTROPIC]-O.bin",
                   "eta-[BAROCLINIC]-A.bin", "rchi-[BAROCLINIC]-O.bin"):
            assert (dirs[k] / fn).read_bytes() == (singles[k] / fn).read_bytes(), (k, fn)
    summary = [ln.split() for ln in (tmp_path / "ensemble_summary.txt").read_text().splitlines() if ln.strip() and ln.split()[0].isdigit()]
    assert [int(r[0]) for r in summary] == list(range(N))
    for k, r in enumerate(summary):
        eff = T.read_efficiency_raw(str(singles[k] / "efficiency.txt"))
        assert float(r[6]) == pytest.approx(eff["eta [L(B=B0dB) = B0]     wo/ boundary"][0], rel=1e-8)      # sum_Qeta of the last variant
        assert float(r[5]) == pytest.approx(eff["sum Q"][0], rel=1e-8)
        iters.add(int(r[1]))
    assert len(iters) > 1, "the cases should stop at different iterations"
    assert not (tmp_path / ".xee_comm_id").exists()


def test_fortran_ensemble_host_two_ranks_on_one_gpu(tmp_path):
    """Two ranks of bin/xee-ensemble on ONE card (XEE_ENSEMBLE_GATHER=files replaces the RCCL gather, which refuses two
    ranks on one device): every rank reads the config FILE named on its command line (a launcher forwards stdin to rank 0
    only), a rendezvous file left behind by a run that died is ignored, each rank writes its own cases' files, rank 0 the
    summary of all -- same bytes as the one-rank run."""
    assert ENSEMBLE.exists()
    nr, nz, N = 80, 56, 5
    mode = ("DYNAMIC_EFFICIENCY", "CYLINDRICAL", "DENSITY_NORMAL", "BAROTROPIC")
    cfg = dict(domain=(0.0, 2.0e5, 0.0, 1.5e4), r1=1e-8, r2=-1.0, alpha=1.0, max_iter=300)
    out = {}
    for world in (1, 2):
        base = tmp_path / f"w{world}"; base.mkdir()
        dirs = []
        for k in range(N):
            A, B, Cc, bc = H.rich_inputs(nr, nz, seed=80 + k)
            d = base / f"case{k}"; d.mkdir()
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("bc_init", bc)):
                np.ascontiguousarray(arr, np.float32).tofile(d / f"{name}.bin")
            dirs.append(d)
        lines = ["-".join(mode), "%.9e %.9e %.9e %.9e" % cfg["domain"], f"{nr} {nz}",
                 "%.9e %.9e %d %.9e" % (cfg["r1"], cfg["r2"], cfg["max_iter"], cfg["alpha"]), str(N)] + [str(d) for d in dirs]
        (base / "ensemble.txt").write_text("\n".join(lines) + "\n")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "XEE_RANK", "XEE_WORLD")}
        if world == 2:
            (base / ".xee_comm_id.job7").write_bytes(b"\0" * 136)        # stale: stamp 0, a dead run's id
            env.update(XEE_WORLD="2", XEE_LOCAL_RANK="0", XEE_ENSEMBLE_GATHER="files", XEE_JOB_ID="job7", XEE_COMM_TIMEOUT_S="60")
        procs = []
        for r in range(world):
            e = dict(env, XEE_RANK=str(r)) if world == 2 else env
            procs.append(subprocess.Popen([str(ENSEMBLE), str(base / "ensemble.txt")], cwd=base, stdin=subprocess.DEVNULL,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e))
        for pr in procs:
            o, _ = pr.communicate(timeout=600)
            assert pr.returncode == 0, o[-3000:]
        out[world] = (dirs, [ln for ln in (base / "ensemble_summary.txt").read_text().splitlines() if ln.split() and ln.split()[0].isdigit()])
        assert not list(base.glob(".xee_comm_id*")), list(base.glob(".xee_comm_id*"))
    assert out[1][1] == out[2][1] and len(out[2][1]) == N
    for a, b in zip(out[1][0], out[2][0]):
        for fn in ("solver_a-sA.bin", "eta-[BAROTROPIC]-A.bin", "rchi-[BAROTROPIC]-O.bin"):
            assert (a / fn).read_bytes() == (b / fn).read_bytes(), (a, fn)


# ---- the legacy program's Fortran host (bin/xee-old-diagnose) -------------------------------------------------------

OLD_HOST = ROOT / "xlab-ee-fortran_amd" / "bin" / "xee-old-diagnose"
RELAX1 = re.compile(r"Relaxation uses\s+(\d+)\s+steps\. Final residue is\s+(\S+)\s*\.")


@pytest.mark.parametrize("name", ["legacy_64x48_tendency", "legacy_64x48_baroall_bc"])
def test_fortran_legacy_host_reproduces_the_legacy_binary(name, tmp_path):
    """`xee-old-diagnose < diag.txt` with the legacy program's own config and input files (plus the `legacy_records` line:
    read Q/F records and truncate the exchange-conversion metric the way the legacy binary does) writes EVERY file of
    oracle/_ref/old-diagnose byte for byte, prints the same iteration counts and the same efficiency.txt numbers."""
    import json
    assert OLD_HOST.exists(), "build it: make -C xlab-ee-fortran_amd fortran"
    g = np.load(H.GOLDEN / f"{name}.npz")
    meta = json.loads(str(g["meta"]))
    (tmp_path / "out").mkdir()
    for k in ("A", "B", "C", "Q", "F"):
        g["in_" + k].tofile(tmp_path / f"{k}.bin")
    lines = [meta["mode"], "%.9e // testing_dt" % meta["testing_dt"], " ".join("%.9e" % v for v in meta["domain"]), "64 48", ".", "out",
             "A.bin", "B.bin", "C.bin", "Q.bin", "F.bin",
             "%d %.9e %d %.9e" % tuple(meta["psi_solver"]), "%d %.9e %d %.9e" % tuple(meta["chi_solver"])]
    for key in ("psi_bc", "chi_bc"):
        if "in_" + key in g.files:
            g["in_" + key].tofile(tmp_path / f"{key}.bin")
            lines += ["yes", f"{key}.bin"]
        else:
            lines.append("no")
    lines.append("legacy_records   // Q, F as the legacy binary really holds them; INTEGER r, dr, dz in the exchange conversion")
    (tmp_path / "diag.txt").write_text("\n".join(lines) + "\n")
    with open(tmp_path / "diag.txt") as fin:
        p = subprocess.run([str(OLD_HOST)], stdin=fin, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    wrote = [k[4:] for k in g.files if k.startswith("out_")]
    for fn in wrote:
        got = np.fromfile(tmp_path / "out" / fn, np.float32)
        assert got.tobytes() == np.ascontiguousarray(g["out_" + fn]).tobytes(), fn
    assert [int(n) for n, _ in RELAX1.findall(p.stdout)] == [it for it, _ in meta["relax"]]
    want = T.read_efficiency_raw_text(meta["efficiency_txt"]) if hasattr(T, "read_efficiency_raw_text") else None
    got_eff = T.read_efficiency_raw(str(tmp_path / "out" / "efficiency.txt"))
    ref_eff = {}
    for line in meta["efficiency_txt"].splitlines():
        if ":" in line and not line.lstrip().startswith("#"):
            label, rest = line.split(":", 1)
            ref_eff[label.strip()] = [float(x) for x in rest.split(",")]
    assert set(got_eff) == set(ref_eff)
    for label, nums in ref_eff.items():
        if label.startswith("Time elapsed"):
            continue
        assert got_eff[label][0] == pytest.approx(nums[0], rel=2e-7, abs=0), label
        if len(nums) > 1:
            assert got_eff[label][1] == pytest.approx(nums[1], rel=4e-7, abs=0), label
    # and the XEffReader-compatible reader sees its six keys
    assert set(T.read_efficiency(str(tmp_path / "out" / "efficiency.txt"))) == {"semi_internal", "internal", "semi_cb1", "wtheta", "local_response", "semi_total"}
