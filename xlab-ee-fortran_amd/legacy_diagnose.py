"""The reference's LEGACY program -- the "energy-efficiency diagnostic sweep" proper -- on the GPU, for a BATCH of cases:
src/old-diagnose/diagnose.f90:283-841 (TENDENCY mode).

    source terms (A11)   sum_Q :284; solver coefficients :289-312; staggered averages :329-354; m2 :359-380 (intended form);
                         J = Q/(Cp exner) :383-387; RHS_rpsi_thm :395-406; RHS_rpsi_mom :412-436
    Stage I              psi for thm + mom on the full-B operator :449-463
    Stage II             u, w :468; dtheta/dt :470-479; anomaly dB, updated rhoA/rhoB, theta :489-518
    Stage III            f_basic, f_anomaly :523-541; up to six chi solves on {L(B=0), L(B0+dB)} x {0 with the chi boundary,
                         -dB, -B0} :548-673, each followed by cal_eta and sum(Q eta)
    integral check       psi solves on both operators, w*theta sums :679-725
    exchange conversion  methods 1 and 2 :730-772
    efficiency.txt       :779-841, every label (readable by xee_tools.read_efficiency / the reference's XEffReader)

Every numerical step is a libxee call (include/xee.h): the solves run as one device-resident ensemble whose members are
the cases of the batch (each stops on its own criterion), the O(N) stencils through the Tier-1 entry points.  This module
is host-side orchestration only; there is no CPU fallback (libxee raises without a GPU).

First guesses.  `chained=True` (default) is the program's own order -- rchi / rpsi are carried from one solve to the next
("the order ... is such that the initial guessing field is better", :545) -- and reproduces the legacy binary's files bit
for bit and its efficiency.txt to the printed digits (tests/test_gpu_legacy_diagnose.py against
tests/golden/legacy_64x48_*.npz).  `chained=False` starts every solve from its own boundary field: the
solves of one stage are then independent.

Solver mapping (old-diagnose/xtt-lib/elliptic_tools.f90:93-302 onto xee_ensemble_solve): strategy 1 == r1 = r, r2 off,
converge_time 1; strategy 2 == r1 off, r2 = r, converge_time 10, lost_rate 5; strategies 3/4 (max norm over the whole
array) are not offered; max_iter must be a multiple of 100.

Defined where the legacy program has undefined behaviour: Q and F are B-cell fields as given (to
reproduce a legacy run pass `sheared(record)`: the program reads nr x nz records into (nr-1,nz-1) arrays); m2 in its
intended form; the "w/ boundary" sums are 0 without a chi boundary file; `legacy_int=True` keeps the program's INTEGER
r, dr, dz in the exchange conversion (False: the documented fix with real metric factors).
"""
import os
import time

import numpy as np

from . import xee as X

f32 = np.float32
G0, THETA0 = f32(9.8), f32(298.0)          # xtt-lib-fortran/constants.f90:4


def sheared(record, nr, nz):
    """What the legacy program holds in Q_in / F_in after reading an nr x nz record into an (nr-1,nz-1) array
    (:211-212 vs :241-242): the first (nr-1)(nz-1) floats of the record, column-major."""
    flat = np.ascontiguousarray(record, f32).reshape(-1)
    return np.ascontiguousarray(flat[: (nr - 1) * (nz - 1)].reshape(nz - 1, nr - 1))


def _solver_args(spec):
    strategy, r, max_iter, alpha = spec
    if max_iter % 100:
        raise ValueError("max_iter must be a multiple of the check step (100), as the legacy loop assumes")
    if strategy == 1:
        return dict(max_iter=int(max_iter), r1=float(r), r2=-1.0, alpha=float(alpha), converge_time=1, lost_rate=5)
    if strategy == 2:
        return dict(max_iter=int(max_iter), r1=-1.0, r2=float(r), alpha=float(alpha), converge_time=10, lost_rate=5)
    raise ValueError("legacy strategies 3 and 4 (max norm over the whole array) are not supported")


def run(cases, *, geometry=0, density_mode=0, domain=(0.0, 1.0, 0.0, 1.0), testing_dt=100.0, operator="BARO_ALL",
        psi_solver=(1, 1e-7, 2000, 1.0), chi_solver=(1, 1e-7, 2000, 1.0), legacy_int=True, chained=True, device=-1,
        first_case_id=0):
    """cases: list of dict(A, B, C (nz,nr) O-grid planes; Q, F (nz-1,nr-1) B cells; optional psi_bc, chi_bc (nz,nr)) --
    all with or all without each boundary field.  Returns a list (one per case) of
    dict(files={legacy file name: array}, scalars, relax=[(iterations, err_now)], lines=[(label, value, ratio)], record)."""
    M = len(cases)
    nz, nr = cases[0]["A"].shape
    has_psi_bc = cases[0].get("psi_bc") is not None
    has_chi_bc = cases[0].get("chi_bc") is not None
    bt = operator in ("BAROTROPIC", "BARO_ALL")
    bcl = operator in ("BAROCLINIC", "BARO_ALL")
    prof = X.profiles(nr, nz, geometry, density_mode, domain)
    sin_t = prof["sin_table"] if geometry == 1 else None
    stack = lambda key: np.stack([np.ascontiguousarray(c[key], f32) for c in cases])
    A, B, Cc, Q, F = (stack(k) for k in ("A", "B", "C", "Q", "F"))
    zero_O = np.zeros((M, nz, nr), f32)
    out = [dict(files={}, scalars={}, relax=[]) for _ in range(M)]
    psi_args, chi_args = _solver_args(psi_solver), _solver_args(chi_solver)

    with X.Ensemble(nr, nz, M, device=device) as ens:
        ens.set_profiles(prof)
        ens.set_case_ids(first_case_id)
        ens.upload("A", A); ens.upload("B", B); ens.upload("C", Cc); ens.upload("Q", Q)
        ens.assemble(X.BAROCLINIC)                      # K1 (:289-312) + the full-B stencil of Stage I (:449)
        sA, sB, sC = ens.download("SOLVER_A"), ens.download("SOLVER_B"), ens.download("SOLVER_C")

        # ---- source terms and the fields Stage II needs, per case (O(N) stencils, Tier 1)
        pre = []
        for m in range(M):
            rhoA_A, rhoB_C, rhoB_B, rhoC_C = X.legacy_averages(A[m], B[m], Cc[m])           # :329-354
            m2 = X.m2_from_C(rhoC_C, prof, sin_t)                                            # :359-380
            JJ = X.JJ_B(Q[m], prof["exner"])                                                 # :383-387
            thm = X.rhs_thermal(Q[m], prof["ra"], prof["exner"])                             # :395-406
            mom = X.rhs_momentum(m2, F[m], prof)                                             # :412-436
            pre.append(dict(rhoA_A=rhoA_A, rhoB_C=rhoB_C, rhoB_B=rhoB_B, JJ=JJ, f_psi=X.add_fields(thm, mom)))
            fl = out[m]["files"]
            fl["J-B.bin"], fl["RHS_rpsi_thm-O.bin"], fl["RHS_rpsi_mom-O.bin"] = JJ, thm, mom
            fl["solver_a-sA.bin"], fl["solver_b-B.bin"], fl["solver_c-sC.bin"] = sA[m], sB[m], sC[m]
            out[m]["scalars"]["sum_Q"] = X.integrate_weight_B(Q[m], prof)                    # :284
        f_psi = np.stack([p["f_psi"] for p in pre])
        psi_start = stack("psi_bc") if has_psi_bc else zero_O

        def solve(args, start=None):
            """one solve_elliptic on every case; start None = carry the current field on (the program's chaining)"""
            a = dict(args)
            if start is not None:
                ens.upload("BC_INIT", start)
            rec = ens.solve(a.pop("max_iter"), a.pop("r1"), a.pop("r2"), a.pop("alpha"),
                            flags=X.SOLVE_DEFAULT if start is not None else X.SOLVE_KEEP_PSI, **a)
            for m in range(M):
                out[m]["relax"].append((int(rec["iters"][m]), f32(rec["err_now"][m])))
            return ens.download("PSI")

        # ---- Stage I (:449-463)
        ens.upload("FORCING", f_psi)
        rpsi = solve(psi_args, psi_start)
        ens.cal_uw()                                                                          # rpsiToUW :915-941
        U, W = ens.download("U"), ens.download("W")
        # ---- Stage II (:468-518), per case
        theta, b_anom, sb_anom = [], [], []
        for m in range(M):
            fl, p = out[m]["files"], pre[m]
            fl["rpsi_before-O.bin"], fl["w_before-A.bin"], fl["u_before-C.bin"] = rpsi[m], W[m], U[m]
            dth = X.theta_tendency(p["JJ"], p["rhoA_A"], W[m], p["rhoB_C"], U[m])             # :470-479
            fl["dtheta_dt-B.bin"] = dth
            out[m]["scalars"]["sum_dtheta_dt"] = X.integrate_weight_B(dth, prof)              # :487
            upd = X.theta_anomaly_update(dth, testing_dt, p["rhoB_B"], p["rhoA_A"], p["rhoB_C"], prof)   # :489-518
            fl["theta_after-B.bin"] = upd["theta"]
            theta.append(upd["theta"]); b_anom.append(upd["b_anomaly"]); sb_anom.append(upd["solver_b_anomaly"])
        # ---- Stage III right-hand sides (:523-541)
        f_basic = np.stack([X.rhs_from_B(pre[m]["rhoB_B"]) for m in range(M)])                # b_basic_B = rhoB_B (:348)
        f_anom = np.stack([X.rhs_from_B(b_anom[m]) for m in range(M)])
        for m in range(M):
            out[m]["files"]["RHS_rchi-O.bin"] = X.add_fields(f_basic[m], f_anom[m])           # :540
        sb_B0dB = np.stack([X.add_fields(sB[m], sb_anom[m]) for m in range(M)])               # :572

        def set_operator(name):
            if name == "0":
                ens.assemble_solver(False)                                                    # solverB_B = 0 (:553)
            else:
                ens.upload("SOLVER_B", sb_B0dB)
                ens.assemble_solver(True)
        ops = (["0"] if bt else []) + (["B0dB"] if bcl else [])

        def chi_solve(op, rhs, f, start):
            set_operator(op)
            ens.upload("FORCING", f)
            rchi = solve(chi_args, start)
            ens.cal_eta()                                                                     # cal_eta :1129-1141
            rec = ens.ee_scalars()                                                            # cal_sum_Qeta :1073-1092
            eta = ens.download("ETA")
            for m in range(M):
                out[m]["scalars"][f"sum_Qeta_{op}_{rhs}"] = f32(rec["sum_Qeta"][m])
                out[m]["files"][f"eta-[{op}_{rhs}]-A.bin"], out[m]["files"][f"rchi-[{op}_{rhs}]-O.bin"] = eta[m], rchi[m]

        if has_chi_bc:                                                                        # :548-588
            chi_start = stack("chi_bc")
            for n, op in enumerate(ops):
                chi_solve(op, "0", zero_O, chi_start if (n == 0 or not chained) else None)
        first = True                                                                          # rchi = 0 (:590)
        for rhs, f in (("dB", f_anom), ("B0", f_basic)):                                      # :594-673
            for op in ops:
                chi_solve(op, rhs, f, zero_O if (first or not chained) else None)
                first = False
        # ---- integral check (:679-725)
        for n, op in enumerate(ops):
            set_operator(op)
            ens.upload("FORCING", f_psi)
            rpsi = solve(psi_args, psi_start if (n == 0 or not chained) else None)
            ens.cal_uw()
            U, W = ens.download("U"), ens.download("W")
            for m in range(M):
                fl = out[m]["files"]
                fl[f"rpsi_after-[{op}]-O.bin"], fl[f"w_after-[{op}]-A.bin"], fl[f"u_after-[{op}]-C.bin"] = rpsi[m], W[m], U[m]
                wth = X.cal_wtheta(W[m], theta[m])                                            # :1117-1127
                fl[f"wtheta_JF_after-[{op}]-B.bin"] = wth
                out[m]["scalars"][f"sum_wtheta_{op}_JF"] = f32(X.integrate_weight_B(wth, prof) * (G0 / THETA0))   # :703
        records = ens.records()
    # ---- exchange conversion, methods 1 and 2 (:730-772)
    if has_chi_bc:
        for m in range(M):
            fl, S = out[m]["files"], out[m]["scalars"]
            for op in ops:
                c0, cdB, cB0 = (fl[f"rchi-[{op}_{r}]-O.bin"] for r in ("0", "dB", "B0"))
                for meth, chi_sum in (("", X.add_fields(X.add_fields(c0, cdB), cB0)), ("2", X.add_fields(cdB, cB0))):
                    bnd, s = X.exchange_conversion(fl[f"rpsi_after-[{op}]-O.bin"], chi_sum, Cc[m], prof, legacy_int=legacy_int)
                    fl[f"bndconv{meth}-[{op}].bin"], S[f"sum_bndconv{meth}_{op}"] = bnd, f32(s)
    for m in range(M):
        out[m]["lines"] = efficiency_lines(out[m]["scalars"], bt, bcl, has_chi_bc)
        out[m]["record"] = records[m]
        out[m]["profiles"] = prof
    return out


def efficiency_lines(S, bt, bcl, has_chi_bc):
    """efficiency.txt (old-diagnose/diagnose.f90:779-841) as [(label, value, ratio | None) | ("#", comment)] in the
    program's order; the scalar arithmetic (ratios, the decomposition sum :805,:833) is fp32 as in the program."""
    sq = f32(S["sum_Q"])
    g = lambda key: f32(S.get(key, 0.0))
    out = [("sum Q                                      ", sq, None), ("sum dtheta_dt                              ", g("sum_dtheta_dt"), None),
           ("Local heat response (sum Q / sum dtheta_dt)", g("sum_dtheta_dt") / sq, None)]      # (:784 divides dtheta by Q)
    for on, op, present in (("B=0)   ", "0", bt), ("B=B0dB)", "B0dB", bcl)):
        if not present:
            continue
        two = lambda lab, v: out.append((lab, f32(v), f32(v) / sq))
        out.append(("#", "Boundary efficiency"))
        if has_chi_bc:
            two(f"eta [L({on} = 0]      w/  boundary", g(f"sum_Qeta_{op}_0"))
        out.append(("#", "Internal efficiency"))
        two(f"eta [L({on} = dB]     wo/ boundary", g(f"sum_Qeta_{op}_dB"))
        two(f"eta [L({on} = B0]     wo/ boundary", g(f"sum_Qeta_{op}_B0"))
        if has_chi_bc:
            out.append(("#", "Boundary conversion (Method 1)"))
            two("bndconv [L(B=0) = B0dB]   w/ boundary" if op == "0" else "bndconv [L(B=B0dB) = B0dB]w/ boundary", g(f"sum_bndconv_{op}"))
            out.append(("#", "Boundary conversion (Method 2)"))
            two("bndconv2 [L(B=0) = B0dB]   w/ boundary" if op == "0" else "bndconv2 [L(B=B0dB) = B0dB]w/ boundary", g(f"sum_bndconv2_{op}"))
        out.append(("#", "Decomposition sum"))
        tmp = f32(f32(g(f"sum_Qeta_{op}_0") + g(f"sum_Qeta_{op}_dB")) + g(f"sum_Qeta_{op}_B0"))
        if has_chi_bc:
            tmp = f32(tmp + g(f"sum_bndconv_{op}"))
        two(f"etaQ [L({on} = J F] w/  boundary", tmp)
        out.append(("#", "wtheta integral"))
        two(f"wtheta [L({on} = J F] w/  boundary", g(f"sum_wtheta_{op}_JF"))
    return out


def write_outputs(result, outdir, elapsed=None):
    """Write one case's files under the legacy names and its efficiency.txt (ES16.9 numbers: an fp32 value round-trips;
    the reference's list-directed output prints 8 digits)."""
    os.makedirs(outdir, exist_ok=True)
    for name, arr in result["files"].items():
        np.ascontiguousarray(arr, f32).tofile(os.path.join(outdir, name))
    with open(os.path.join(outdir, "efficiency.txt"), "w") as fh:
        fh.write(" Time elapsed (sec)                          : %16.9E\n" % (0.0 if elapsed is None else elapsed))
        for item in result["lines"]:
            if item[0] == "#":
                fh.write(" # %s\n" % item[1])
            elif item[2] is None:
                fh.write(" %s : %16.9E\n" % (item[0], float(item[1])))
            else:
                fh.write(" %s : %16.9E , %16.9E\n" % (item[0], float(item[1]), float(item[2])))
    return os.path.join(outdir, "efficiency.txt")


def run_and_write(cases, outdirs, **kw):
    t0 = time.perf_counter()
    res = run(cases, **kw)
    dt = time.perf_counter() - t0
    return [write_outputs(r, d, elapsed=dt) for r, d in zip(res, outdirs)], res
