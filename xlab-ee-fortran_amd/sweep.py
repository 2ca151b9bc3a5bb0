"""The source-term sweep (BASELINE.json configs[4], SURVEY.md 8d "Config 5"): for every timestep t
the well-defined part of the legacy program's solve list (src/old-diagnose/diagnose.f90:548-673),
every solve independent and started from its own bc_init:

    operator  in { L(A, B=0, C)  [BAROTROPIC],  L(A, B_t, C)  [BAROCLINIC] }        diagnose.f90:553,572
  x RHS       in { -B_t                          (f = -rhoB_in,   current program, diagnose.f90:2)
                   thermal  g0/theta0 d/dr(Q/(Cp exner))          (old-diagnose :383-406)
                   zero RHS with chi = 1e-5 sin(pi z) on the r = Lr2 rim  (old-diagnose :548-568) }

= 6 cases per timestep, case id = 6*t + 3*operator + rhs.  Each case yields eta and the EE scalars
sum_Q, sum_Qeta, efficiency (old-diagnose :1029-1092, :792-793) in its 128-byte record.

The cases of this rank's timesteps are the members of one GPU batch (class Sweep: the three kinds of a
timestep share their operator and are solved together); ranks own blocks of timesteps
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


class Sweep:
    """This rank's block of timesteps, resident on the GPU.  __init__ uploads every input; step() runs the stages and
    touches no host field (bench.py times it).

    fused=True (default): ONE batch of 3 x n_timesteps members -- member rhs * n_timesteps + k is right-hand-side kind rhs
    of local timestep k -- so that the three kinds of a timestep, which share their operator, are solved together: two
    stages (one per operator) of 3 n cases each fill the GPU better than six stages of n.  The boundary field never
    changes (zero for the first two kinds, the rim field for the third) and the forcing is rebuilt per stage through the
    member window.  fused=False runs the six (operator, kind) stages one after the other on n_timesteps members: the same
    results, and per-kind timing (stage_timing) for diagnostics."""

    def __init__(self, nr, nz, timesteps, *, device=-1, domain=(0.0, 1.0, 0.0, 1.0), options=None, fused=True):
        self.nr, self.nz, self.timesteps, self.fused = nr, nz, list(timesteps), fused
        n = len(self.timesteps)
        self.ens = X.Ensemble(nr, nz, 3 * n if fused else n, device=device)
        ens = self.ens
        for k, v in (options or {}).items():
            ens.set_option(k, v)
        ens.set_profiles(X.profiles(nr, nz, 0, 0, domain))
        Q = heating_Q(nr, nz)
        rim = rim_bc(nr, nz)
        zero = np.zeros((nz, nr), np.float32)
        for k, t in enumerate(self.timesteps):
            A, B, Cc, _ = ensemble_member(nr, nz, t)
            for rhs in (range(3) if fused else (0,)):
                m = rhs * n + k
                ens.upload("A", A, first=m); ens.upload("B", B, first=m); ens.upload("C", Cc, first=m)
                ens.upload("Q", Q, first=m)
                ens.upload("BC_INIT", rim if (fused and rhs == 2) else zero, first=m)
        if not fused:
            ens.field_save("BC_INIT", 0)                              # slot 0: zero first guess and boundary
            for k in range(n):
                ens.upload("BC_INIT", rim, first=k)
            ens.field_save("BC_INIT", 1)                              # slot 1: the rim field (old-diagnose :548)
        self.stage_timing = {}

    def close(self):
        self.ens.close()

    def _forcing(self, rhs):
        ens = self.ens
        if rhs == 0:
            ens.forcing_from_B()                                      # f = -rhoB_in (diagnose.f90:2)
        elif rhs == 1:
            ens.forcing_thermal()                                     # old-diagnose :383-406
        else:
            ens.forcing_zero()                                        # old-diagnose :552

    def step(self, max_iter=100000, r1=1e-8, r2=-1.0, alpha=1.0, on_stage=None):
        """-> records of this rank's cases, ordered (local timestep, operator, rhs).  on_stage(op, rhs, ens, first) is
        called after each stage's EE scalars with the first member of kind rhs (tests download fields there); in fused
        mode once per kind of the stage."""
        ens = self.ens
        n = len(self.timesteps)
        recs = np.zeros(n * CASES_PER_TIMESTEP, X.RECORD_DTYPE)

        def take(rec, op, rhs, first):
            for k, t in enumerate(self.timesteps):
                slot = CASES_PER_TIMESTEP * k + 3 * op + rhs
                recs[slot] = rec[first + k]
                recs[slot]["case_id"] = CASES_PER_TIMESTEP * t + 3 * op + rhs

        for op in (X.BAROTROPIC, X.BAROCLINIC):
            ens.assemble(op)
            if self.fused:
                for rhs in range(3):
                    ens.member_window(rhs * n, n)
                    self._forcing(rhs)
                ens.member_window()
                ens.solve(max_iter, r1, r2, alpha)
                self.stage_timing[(OP_NAMES[op], "all three kinds")] = ens.timing()
                ens.cal_eta()
                rec = ens.ee_scalars()
                for rhs in range(3):
                    take(rec, op, rhs, rhs * n)
                    if on_stage is not None:
                        on_stage(op, rhs, ens, rhs * n)
            else:
                for rhs, name in enumerate(RHS_NAMES):
                    self._forcing(rhs)
                    ens.field_restore("BC_INIT", 1 if name == "rim_bc" else 0)
                    ens.solve(max_iter, r1, r2, alpha)
                    self.stage_timing[(OP_NAMES[op], name)] = ens.timing()
                    ens.cal_eta()
                    rec = ens.ee_scalars()
                    take(rec, op, rhs, 0)
                    if on_stage is not None:
                        on_stage(op, rhs, ens, 0)
        return recs


def run(nr, nz, n_timesteps, *, rank=0, world=1, max_iter=100000, r1=1e-8, r2=-1.0, alpha=1.0,
        domain=(0.0, 1.0, 0.0, 1.0), device=-1, gather_device=None, keep_fields=False, comm=None, fused=True):
    """Run this rank's block of timesteps; returns (all_records_sorted_by_case_id, local_fields).
    local_fields[(t, op, rhs)] = dict(psi=..., eta=...) only when keep_fields."""
    lo, hi = dp.partition(n_timesteps, world, rank)
    fields = {}
    recs = np.zeros(0, X.RECORD_DTYPE)
    if hi > lo:
        sw = Sweep(nr, nz, range(lo, hi), device=device, domain=domain, fused=fused)
        try:
            def keep(op, rhs, ens, first):
                psi, eta = ens.download("PSI", first, hi - lo), ens.download("ETA", first, hi - lo)
                for k, t in enumerate(range(lo, hi)):
                    fields[(t, op, rhs)] = {"psi": psi[k], "eta": eta[k]}
            recs = sw.step(max_iter, r1, r2, alpha, on_stage=keep if keep_fields else None)
        finally:
            sw.close()
    # records are ordered by timestep block, i.e. by case id, on every rank
    per = -(-n_timesteps // world) * CASES_PER_TIMESTEP
    allrec = dp.gather_records(recs, n_timesteps * CASES_PER_TIMESTEP, world, rank, device=gather_device,
                               per_rank=per, comm=comm) if world > 1 else recs
    assert len(allrec) == n_timesteps * CASES_PER_TIMESTEP
    return allrec, fields
