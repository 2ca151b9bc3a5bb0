"""The source-term sweep (BASELINE.json configs[4], SURVEY.md 8d "Config 5"): for every timestep t
the well-defined part of the legacy program's solve list (src/old-diagnose/diagnose.f90:548-673),
every solve independent and started from its own bc_init:

    operator  in { L(A, B=0, C)  [BAROTROPIC],  L(A, B_t, C)  [BAROCLINIC] }        diagnose.f90:553,572
  x RHS       in { -B_t                          (f = -rhoB_in,   current program, diagnose.f90:2)
                   thermal  g0/theta0 d/dr(Q/(Cp exner))          (old-diagnose :383-406)
                   zero RHS with chi = 1e-5 sin(pi z) on the r = Lr2 rim  (old-diagnose :548-568) }

= 6 cases per timestep, case id = 6*t + 3*operator + rhs.  Each case yields eta and the EE scalars
sum_Q, sum_Qeta, efficiency (old-diagnose :1029-1092, :792-793) in its 128-byte record.

Timesteps are the ensemble members of one GPU batch (`Ensemble(nr, nz, n_local_timesteps)`): a
solve stage runs for all local timesteps at once; ranks own blocks of timesteps
(ensemble_dp.partition) and only exchange the records (ensemble_dp.gather_records).
Host-side orchestration only -- every numerical stage is a libxee call.
"""
import numpy as np

from . import ensemble_dp as dp
from . import xee as X
from .xee_tools import ensemble_member

RHS_NAMES = ("minus_B", "thermal", "rim_bc")
OP_NAMES = ("BAROTROPIC", "BAROCLINIC")
CASES_PER_TIMESTEP = 6


def heating_Q(nr, nz):
    """Q = exp(-((r-0.3)/0.1)^2 - ((z-0.5)/0.2)^2) on B cells (cell centres), float64 math -> f32."""
    r = np.linspace(0.0, 1.0, nr)
    z = np.linspace(0.0, 1.0, nz)
    rb = 0.5 * (r[1:] + r[:-1])[None, :]
    zb = 0.5 * (z[1:] + z[:-1])[:, None]
    return np.exp(-((rb - 0.3) / 0.1) ** 2 - ((zb - 0.5) / 0.2) ** 2).astype(np.float32)


def rim_bc(nr, nz):
    """bc_init = 1e-5 sin(pi z) on the outer-radius rim, zero elsewhere."""
    z = np.linspace(0.0, 1.0, nz)
    bc = np.zeros((nz, nr), np.float32)
    bc[:, -1] = (1e-5 * np.sin(np.pi * z)).astype(np.float32)
    return bc


def run(nr, nz, n_timesteps, *, rank=0, world=1, max_iter=100000, r1=1e-8, r2=-1.0, alpha=1.0,
        domain=(0.0, 1.0, 0.0, 1.0), device=-1, gather_device=None, keep_fields=False):
    """Run this rank's block of timesteps; returns (all_records_sorted_by_case_id, local_fields).
    local_fields[(t, op, rhs)] = dict(psi=..., eta=...) only when keep_fields."""
    lo, hi = dp.partition(n_timesteps, world, rank)
    n_local = hi - lo
    prof = X.profiles(nr, nz, 0, 0, domain)
    Q = heating_Q(nr, nz)
    zero_bc = np.zeros((nz, nr), np.float32)
    rim = rim_bc(nr, nz)
    fields = {}
    recs = np.zeros(n_local * CASES_PER_TIMESTEP, X.RECORD_DTYPE)
    if n_local > 0:
        with X.Ensemble(nr, nz, n_local, device=device) as ens:
            ens.set_profiles(prof)
            for k, t in enumerate(range(lo, hi)):
                A, B, Cc, _ = ensemble_member(nr, nz, t)
                ens.upload("A", A, first=k); ens.upload("B", B, first=k); ens.upload("C", Cc, first=k)
                ens.upload("Q", Q, first=k)
            for op in (X.BAROTROPIC, X.BAROCLINIC):
                ens.assemble(op)
                for rhs, name in enumerate(RHS_NAMES):
                    if name == "minus_B":
                        ens.forcing_from_B()
                    elif name == "thermal":
                        ens.forcing_thermal()
                    else:
                        ens.forcing_zero()
                    bc = rim if name == "rim_bc" else zero_bc
                    for k in range(n_local):
                        ens.upload("BC_INIT", bc, first=k)
                    ens.solve(max_iter, r1, r2, alpha)
                    ens.cal_eta()
                    rec = ens.ee_scalars()
                    for k, t in enumerate(range(lo, hi)):
                        cid = CASES_PER_TIMESTEP * t + 3 * op + rhs
                        slot = CASES_PER_TIMESTEP * k + 3 * op + rhs
                        recs[slot] = rec[k]
                        recs[slot]["case_id"] = cid
                    if keep_fields:
                        psi, eta = ens.download("PSI"), ens.download("ETA")
                        for k, t in enumerate(range(lo, hi)):
                            fields[(t, op, rhs)] = {"psi": psi[k], "eta": eta[k]}
    # records are ordered by timestep block, i.e. by case id, on every rank
    per = -(-n_timesteps // world) * CASES_PER_TIMESTEP
    allrec = dp.gather_records(recs, n_timesteps * CASES_PER_TIMESTEP, world, rank, device=gather_device,
                               per_rank=per) if world > 1 else recs
    assert len(allrec) == n_timesteps * CASES_PER_TIMESTEP
    return allrec, fields
