"""PMC workload (developer tool): the sweep of one bench.py workload plus two kernels of KNOWN byte count
that calibrate FETCH_SIZE for the two load widths used (k_negate: 16 B/lane; k_cal_eta: 4 B/lane).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/pmc_workload.py <workload>

workload: ensemble | baroclinic | single | highres   (the shapes of bench.py --workload) [iterations [option=value ...]]
Runs 200 iterations (two blocks of 12 eight-iteration launches plus one residual-writing launch at iterations 100 and 200),
so every launch is the production kernel on production data; tools/pmc_summarize.py turns the CSVs into
profiles/<round>/pmc_summary.json and profiles/traffic.json.
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (member_fields: the bench's own synthetic members)
import xee_pkg  # noqa: E402

SHAPES = {"ensemble": (1024, 512, 32, 0), "baroclinic": (1024, 512, 32, 1), "single": (1024, 512, 1, 0),
          "highres": (4096, 2048, 1, 0)}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "ensemble"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[3:])      # e.g. tile=416 tb_T=10
    nr, nz, M, variant = SHAPES[wl]
    xee = xee_pkg.load()
    with xee.Ensemble(nr, nz, M) as e:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_profiles(xee.profiles(nr, nz))
        for m in range(M):
            A, B, Cc, bc, Q = bench.member_fields(nr, nz, m)
            for name, arr in (("A", A), ("B", B), ("C", Cc), ("BC_INIT", bc), ("Q", Q)):
                e.upload(name, arr, first=m)
        e.forcing_from_B()                 # k_negate: float4 copy of M planes (known bytes, 16 B/lane)
        e.assemble(variant)
        e.solve(iters, 1e-8, -1.0, 1.0)
        print("shared_planes=%d" % e.timing()["shared_planes"], flush=True)   # (read by tools/pmc_summarize.py from the pass's log)
        e.cal_eta()                        # k_cal_eta: dword reads of M planes (known bytes, 4 B/lane)
        e.ee_scalars()


if __name__ == "__main__":
    main()
