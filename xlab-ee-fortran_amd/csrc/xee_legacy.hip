// xee_legacy.hip -- the LEGACY program's stage order, once: src/old-diagnose/diagnose.f90:283-841 (TENDENCY mode) for a batch
// of cases, behind the C-ABI (xee_legacy_run and friends, include/xee.h).  Both hosts call this: the Fortran program
// bin/xee-old-diagnose (one case, the legacy config and files) and the Python batch driver legacy_diagnose.py.
//
//     source terms (A11)   sum_Q :284; solver coefficients :289-312; staggered averages :329-354; m2 :359-380 (intended form);
//                          J = Q/(Cp exner) :383-387; RHS_rpsi_thm :395-406; RHS_rpsi_mom :412-436
//     Stage I              psi for thm + mom on the full-B operator :449-463
//     Stage II             u, w :468; dtheta/dt :470-479; anomaly dB, updated rhoA/rhoB, theta :489-518
//     Stage III            f_basic, f_anomaly :523-541; up to six chi solves on {L(B=0), L(B0+dB)} x {0 with the chi boundary,
//                          -dB, -B0} :548-673, each followed by cal_eta and sum(Q eta)
//     integral check       psi solves on both operators, w*theta sums :679-725
//     exchange conversion  methods 1 and 2 :730-772
//     efficiency.txt       :779-841, every label (readable by xee_tools.read_efficiency / the reference's XEffReader)
//
// Host-side orchestration only: the solves run as one device-resident ensemble whose members are the cases (each stops on
// its own criterion), the O(N) stencils through the Tier-1 entry points.  No arithmetic of its own besides the program's
// fp32 scalar lines (ratios, the decomposition sum :805,:833, g0/theta0 :703).
#include "../../include/xee.h"
#include "xee_device.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace {

using Field = std::vector<float>;

struct Case {
    std::map<std::string, Field> files;
    std::map<std::string, float> scalars;
    std::vector<std::pair<int, float>> relax;        // (iterations, err_now) of every solve, in the program's order
    xee_case_record_t record{};
};

struct Line { std::string label; float value = 0.f, ratio = 0.f; int kind = 0; };   // kind 0: "# comment", 1: value, 2: value , ratio

}  // namespace

struct xee_legacy {
    int nr = 0, nz = 0;
    bool bt = false, bcl = false, has_chi_bc = false;
    std::vector<Case> cases;
    std::vector<std::string> field_order, scalar_order;                  // names in the order they were produced
    std::map<std::string, std::pair<int, int>> shape;                    // file name -> (n1, n2), n1 fastest
};

namespace {

#define LCHK(expr)                       \
    do {                                 \
        int s_ = (expr);                 \
        if (s_ != XEE_OK) return s_;     \
    } while (0)

int bad(const char* msg) { return xee::record_failure(XEE_ERR_INVALID_ARGUMENT, msg); }

// strategy 1 == r1 = r, r2 off, converge_time 1; strategy 2 == r1 off, r2 = r, converge_time 10
// (old-diagnose/xtt-lib/elliptic_tools.f90:93-302 onto xee_ensemble_solve)
struct SolverArgs { int max_iter, converge_time; float r1, r2, alpha; };
int solver_args(int strategy, float r, int max_iter, float alpha, SolverArgs* out)
{
    if (max_iter % 100) return bad("legacy solver line: max_iter must be a multiple of the check step (100), as the legacy loop assumes");
    if (strategy == 1) *out = SolverArgs{max_iter, 1, r, -1.0f, alpha};
    else if (strategy == 2) *out = SolverArgs{max_iter, 10, -1.0f, r, alpha};
    else return bad("legacy strategies 3 and 4 (max norm over the whole array) are not supported");
    return XEE_OK;
}

void put(xee_legacy* L, int m, const std::string& name, const float* data, int n1, int n2)
{
    if (!L->shape.count(name)) { L->shape[name] = {n1, n2}; L->field_order.push_back(name); }
    L->cases[m].files[name].assign(data, data + (size_t)n1 * n2);
}
void put_scalar(xee_legacy* L, int m, const std::string& key, float v)
{
    bool known = false;
    for (const std::string& k : L->scalar_order) known = known || k == key;
    if (!known) L->scalar_order.push_back(key);
    L->cases[m].scalars[key] = v;
}
float scalar_or_zero(const Case& c, const std::string& key)
{
    auto it = c.scalars.find(key);
    return it == c.scalars.end() ? 0.0f : it->second;
}

// efficiency.txt (old-diagnose/diagnose.f90:779-841) in the program's order
std::vector<Line> efficiency_lines(const xee_legacy* L, const Case& c)
{
    std::vector<Line> out;
    const float sq = scalar_or_zero(c, "sum_Q");
    auto one = [&](const std::string& lab, float v) { Line l; l.label = lab; l.value = v; l.kind = 1; out.push_back(l); };
    auto two = [&](const std::string& lab, float v) { Line l; l.label = lab; l.value = v; l.ratio = v / sq; l.kind = 2; out.push_back(l); };
    auto note = [&](const std::string& text) { Line l; l.label = text; out.push_back(l); };
    one("sum Q                                      ", sq);
    one("sum dtheta_dt                              ", scalar_or_zero(c, "sum_dtheta_dt"));
    one("Local heat response (sum Q / sum dtheta_dt)", scalar_or_zero(c, "sum_dtheta_dt") / sq);      // (:784 divides dtheta by Q)
    for (int k = 0; k < 2; ++k) {
        const bool present = k == 0 ? L->bt : L->bcl;
        if (!present) continue;
        const std::string on = k == 0 ? "B=0)   " : "B=B0dB)", op = k == 0 ? "0" : "B0dB";
        note("Boundary efficiency");
        if (L->has_chi_bc) two("eta [L(" + on + " = 0]      w/  boundary", scalar_or_zero(c, "sum_Qeta_" + op + "_0"));
        note("Internal efficiency");
        two("eta [L(" + on + " = dB]     wo/ boundary", scalar_or_zero(c, "sum_Qeta_" + op + "_dB"));
        two("eta [L(" + on + " = B0]     wo/ boundary", scalar_or_zero(c, "sum_Qeta_" + op + "_B0"));
        if (L->has_chi_bc) {
            note("Boundary conversion (Method 1)");
            two(k == 0 ? "bndconv [L(B=0) = B0dB]   w/ boundary" : "bndconv [L(B=B0dB) = B0dB]w/ boundary", scalar_or_zero(c, "sum_bndconv_" + op));
            note("Boundary conversion (Method 2)");
            two(k == 0 ? "bndconv2 [L(B=0) = B0dB]   w/ boundary" : "bndconv2 [L(B=B0dB) = B0dB]w/ boundary", scalar_or_zero(c, "sum_bndconv2_" + op));
        }
        note("Decomposition sum");
        float tmp = scalar_or_zero(c, "sum_Qeta_" + op + "_0") + scalar_or_zero(c, "sum_Qeta_" + op + "_dB");   // :805, :833, fp32 left to right
        tmp = tmp + scalar_or_zero(c, "sum_Qeta_" + op + "_B0");
        if (L->has_chi_bc) tmp = tmp + scalar_or_zero(c, "sum_bndconv_" + op);
        two("etaQ [L(" + on + " = J F] w/  boundary", tmp);
        note("wtheta integral");
        two("wtheta [L(" + on + " = J F] w/  boundary", scalar_or_zero(c, "sum_wtheta_" + op + "_JF"));
    }
    return out;
}

// ES16.9 as Fortran prints it (an fp32 value survives the round trip; the reference's list-directed output has 8 digits)
std::string es16_9(float v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%16.9E", (double)v);
    return buf;
}

std::string efficiency_text(const xee_legacy* L, const Case& c, double elapsed)
{
    std::string t = " Time elapsed (sec)                          : " + es16_9((float)elapsed) + "\n";
    for (const Line& l : efficiency_lines(L, c)) {
        if (l.kind == 0) t += " # " + l.label + "\n";
        else if (l.kind == 1) t += " " + l.label + " : " + es16_9(l.value) + "\n";
        else t += " " + l.label + " : " + es16_9(l.value) + " , " + es16_9(l.ratio) + "\n";
    }
    return t;
}

int run(xee_legacy* L, const xee_legacy_config_t& cfg, int nr, int nz, int M, const float* A, const float* B, const float* C,
        const float* Q, const float* F, const float* psi_bc, const float* chi_bc, xee_ensemble_t* ens)
{
    const size_t nO = (size_t)nr * nz, nB = (size_t)(nr - 1) * (nz - 1), nA = (size_t)(nr - 1) * nz, nC = (size_t)nr * (nz - 1);
    const size_t n_sA = (size_t)(nr - 1) * (nz - 2), n_sC = (size_t)(nr - 2) * (nz - 1);
    const float g0 = 9.8f, theta0 = 298.0f;                              // xtt-lib-fortran/constants.f90:4
    SolverArgs psi_args, chi_args;
    LCHK(solver_args(cfg.psi_strategy, cfg.psi_r, cfg.psi_max_iter, cfg.psi_alpha, &psi_args));
    LCHK(solver_args(cfg.chi_strategy, cfg.chi_r, cfg.chi_max_iter, cfg.chi_alpha, &chi_args));

    Field ra(nr), za(nz), rcuva(nr), exner(nz), rho(nz), sin_table(nr);
    float dr = 0.f, dz = 0.f;
    LCHK(xee_profiles(cfg.geometry, cfg.density_mode, cfg.domain, nr, nz, ra.data(), za.data(), rcuva.data(), exner.data(), rho.data(),
                      sin_table.data(), &dr, &dz));
    const float* sin_t = cfg.geometry == 1 ? sin_table.data() : nullptr;
    LCHK(xee_ensemble_set_profiles(ens, ra.data(), za.data(), rcuva.data(), rho.data(), exner.data(), dr, dz));
    LCHK(xee_ensemble_set_case_ids(ens, cfg.first_case_id));
    LCHK(xee_ensemble_upload(ens, XEE_F_A, 0, M, A));
    LCHK(xee_ensemble_upload(ens, XEE_F_B, 0, M, B));
    LCHK(xee_ensemble_upload(ens, XEE_F_C, 0, M, C));
    LCHK(xee_ensemble_upload(ens, XEE_F_Q, 0, M, Q));
    LCHK(xee_ensemble_assemble(ens, XEE_BAROCLINIC));                    // K1 (:289-312) + the full-B stencil of Stage I (:449)
    Field sA(n_sA * M), sB(nB * M), sC(n_sC * M);
    LCHK(xee_ensemble_download(ens, XEE_F_SOLVER_A, 0, M, sA.data()));
    LCHK(xee_ensemble_download(ens, XEE_F_SOLVER_B, 0, M, sB.data()));
    LCHK(xee_ensemble_download(ens, XEE_F_SOLVER_C, 0, M, sC.data()));

    // ---- source terms and the fields Stage II needs, per case (O(N) stencils, Tier 1)
    Field rhoA_A(nA * M), rhoB_C(nC * M), rhoB_B(nB * M), rhoC_C(nC), JJ(nB * M), f_psi(nO * M), m2(nB), thm(nO), mom(nO);
    for (int m = 0; m < M; ++m) {
        LCHK(xee_legacy_averages(A + m * nO, B + m * nO, C + m * nO, &rhoA_A[m * nA], &rhoB_C[m * nC], &rhoB_B[m * nB], rhoC_C.data(), nr, nz));   // :329-354
        LCHK(xee_m2_from_C(rhoC_C.data(), ra.data(), rcuva.data(), sin_t, m2.data(), nr, nz));                    // :359-380
        LCHK(xee_JJ_B(Q + m * nB, exner.data(), &JJ[m * nB], nr, nz));                                              // :383-387
        LCHK(xee_rhs_thermal(Q + m * nB, ra.data(), exner.data(), thm.data(), nr, nz));                             // :395-406
        LCHK(xee_rhs_momentum(m2.data(), F + m * nB, za.data(), rcuva.data(), mom.data(), nr, nz));                 // :412-436
        LCHK(xee_add_fields(thm.data(), mom.data(), &f_psi[m * nO], nO));                                           // :456
        put(L, m, "J-B.bin", &JJ[m * nB], nr - 1, nz - 1);
        put(L, m, "RHS_rpsi_thm-O.bin", thm.data(), nr, nz);
        put(L, m, "RHS_rpsi_mom-O.bin", mom.data(), nr, nz);
        put(L, m, "solver_a-sA.bin", &sA[m * n_sA], nr - 1, nz - 2);
        put(L, m, "solver_b-B.bin", &sB[m * nB], nr - 1, nz - 1);
        put(L, m, "solver_c-sC.bin", &sC[m * n_sC], nr - 2, nz - 1);
        float sq = 0.f;
        LCHK(xee_integrate_weight_B(Q + m * nB, ra.data(), za.data(), rcuva.data(), rho.data(), nr, nz, &sq));      // :284
        put_scalar(L, m, "sum_Q", sq);
    }
    const Field zero_O(nO * M, 0.0f);
    const float* psi_start = psi_bc ? psi_bc : zero_O.data();
    Field field_O(nO * M), U(nC * M), W(nA * M), ETA(nA * M);
    std::vector<xee_case_record_t> rec(M);

    // one solve_elliptic on every case; start == nullptr carries the current field on (the program's chaining, :545)
    auto solve = [&](const SolverArgs& a, const float* start, Field& result) -> int {
        if (start) LCHK(xee_ensemble_upload(ens, XEE_F_BC_INIT, 0, M, start));
        LCHK(xee_ensemble_solve(ens, a.max_iter, 100, a.converge_time, 5, a.r1, a.r2, a.alpha, start ? XEE_SOLVE_DEFAULT : XEE_SOLVE_KEEP_PSI));
        LCHK(xee_ensemble_records(ens, rec.data()));
        for (int m = 0; m < M; ++m) L->cases[m].relax.emplace_back(rec[m].iters, rec[m].err_now);
        return xee_ensemble_download(ens, XEE_F_PSI, 0, M, result.data());
    };

    // ---- Stage I (:449-463)
    LCHK(xee_ensemble_upload(ens, XEE_F_FORCING, 0, M, f_psi.data()));
    LCHK(solve(psi_args, psi_start, field_O));
    LCHK(xee_ensemble_cal_uw(ens));                                                                                 // rpsiToUW :915-941
    LCHK(xee_ensemble_download(ens, XEE_F_U, 0, M, U.data()));
    LCHK(xee_ensemble_download(ens, XEE_F_W, 0, M, W.data()));
    // ---- Stage II (:468-518), per case
    Field theta(nB * M), b_anom(nB * M), sb_anom(nB), sb_B0dB(nB * M), f_basic(nO * M), f_anom(nO * M), tmpO(nO);
    for (int m = 0; m < M; ++m) {
        put(L, m, "rpsi_before-O.bin", &field_O[m * nO], nr, nz);
        put(L, m, "w_before-A.bin", &W[m * nA], nr - 1, nz);
        put(L, m, "u_before-C.bin", &U[m * nC], nr, nz - 1);
        float* th = &theta[m * nB];
        LCHK(xee_theta_tendency(&JJ[m * nB], &rhoA_A[m * nA], &W[m * nA], &rhoB_C[m * nC], &U[m * nC], th, nr, nz));   // :470-479
        put(L, m, "dtheta_dt-B.bin", th, nr - 1, nz - 1);
        float s = 0.f;
        LCHK(xee_integrate_weight_B(th, ra.data(), za.data(), rcuva.data(), rho.data(), nr, nz, &s));               // :487
        put_scalar(L, m, "sum_dtheta_dt", s);
        // Stage III right-hand sides: b_basic_B = rhoB_B BEFORE the update (:348, :523-538)
        LCHK(xee_rhs_from_B(&rhoB_B[m * nB], &f_basic[m * nO], nr, nz));
        LCHK(xee_theta_anomaly_update(th, cfg.testing_dt, &rhoB_B[m * nB], &rhoA_A[m * nA], &rhoB_C[m * nC], &b_anom[m * nB], sb_anom.data(),
                                      ra.data(), za.data(), rcuva.data(), rho.data(), nr, nz));                     // :489-518
        put(L, m, "theta_after-B.bin", th, nr - 1, nz - 1);
        LCHK(xee_rhs_from_B(&b_anom[m * nB], &f_anom[m * nO], nr, nz));
        LCHK(xee_add_fields(&f_basic[m * nO], &f_anom[m * nO], tmpO.data(), nO));                                   // :540
        put(L, m, "RHS_rchi-O.bin", tmpO.data(), nr, nz);
        LCHK(xee_add_fields(&sB[m * nB], sb_anom.data(), &sb_B0dB[m * nB], nB));                                    // :572
    }
    std::vector<std::string> ops;
    if (L->bt) ops.push_back("0");
    if (L->bcl) ops.push_back("B0dB");
    auto set_operator = [&](const std::string& op) -> int {
        if (op == "0") return xee_ensemble_assemble_solver(ens, 0);                                                 // solverB_B = 0 (:553)
        LCHK(xee_ensemble_upload(ens, XEE_F_SOLVER_B, 0, M, sb_B0dB.data()));
        return xee_ensemble_assemble_solver(ens, 1);
    };
    auto chi_solve = [&](const std::string& op, const std::string& rhs, const float* f, const float* start) -> int {
        LCHK(set_operator(op));
        LCHK(xee_ensemble_upload(ens, XEE_F_FORCING, 0, M, f));
        LCHK(solve(chi_args, start, field_O));
        LCHK(xee_ensemble_cal_eta(ens));                                                                            // cal_eta :1129-1141
        LCHK(xee_ensemble_ee_scalars(ens));                                                                         // cal_sum_Qeta :1073-1092
        LCHK(xee_ensemble_records(ens, rec.data()));
        LCHK(xee_ensemble_download(ens, XEE_F_ETA, 0, M, ETA.data()));
        for (int m = 0; m < M; ++m) {
            put_scalar(L, m, "sum_Qeta_" + op + "_" + rhs, rec[m].sum_Qeta);
            put(L, m, "eta-[" + op + "_" + rhs + "]-A.bin", &ETA[m * nA], nr - 1, nz);
            put(L, m, "rchi-[" + op + "_" + rhs + "]-O.bin", &field_O[m * nO], nr, nz);
        }
        return XEE_OK;
    };
    const bool chained = cfg.chained != 0;
    if (chi_bc)                                                                                                     // :548-588
        for (size_t n = 0; n < ops.size(); ++n) LCHK(chi_solve(ops[n], "0", zero_O.data(), (n == 0 || !chained) ? chi_bc : nullptr));
    bool first = true;                                                                                              // rchi = 0 (:590)
    for (int r = 0; r < 2; ++r)                                                                                     // :594-673
        for (const std::string& op : ops) {
            LCHK(chi_solve(op, r == 0 ? "dB" : "B0", r == 0 ? f_anom.data() : f_basic.data(), (first || !chained) ? zero_O.data() : nullptr));
            first = false;
        }
    // ---- integral check (:679-725)
    Field wth(nB);
    for (size_t n = 0; n < ops.size(); ++n) {
        const std::string& op = ops[n];
        LCHK(set_operator(op));
        LCHK(xee_ensemble_upload(ens, XEE_F_FORCING, 0, M, f_psi.data()));
        LCHK(solve(psi_args, (n == 0 || !chained) ? psi_start : nullptr, field_O));
        LCHK(xee_ensemble_cal_uw(ens));
        LCHK(xee_ensemble_download(ens, XEE_F_U, 0, M, U.data()));
        LCHK(xee_ensemble_download(ens, XEE_F_W, 0, M, W.data()));
        for (int m = 0; m < M; ++m) {
            put(L, m, "rpsi_after-[" + op + "]-O.bin", &field_O[m * nO], nr, nz);
            put(L, m, "w_after-[" + op + "]-A.bin", &W[m * nA], nr - 1, nz);
            put(L, m, "u_after-[" + op + "]-C.bin", &U[m * nC], nr, nz - 1);
            LCHK(xee_cal_wtheta(&W[m * nA], &theta[m * nB], wth.data(), nr, nz));                                   // :1117-1127
            put(L, m, "wtheta_JF_after-[" + op + "]-B.bin", wth.data(), nr - 1, nz - 1);
            float s = 0.f;
            LCHK(xee_integrate_weight_B(wth.data(), ra.data(), za.data(), rcuva.data(), rho.data(), nr, nz, &s));   // :1094-1113
            put_scalar(L, m, "sum_wtheta_" + op + "_JF", s * (g0 / theta0));                                        // :703
        }
    }
    LCHK(xee_ensemble_records(ens, rec.data()));
    for (int m = 0; m < M; ++m) L->cases[m].record = rec[m];
    // ---- exchange conversion, methods 1 and 2 (:730-772)
    if (chi_bc) {
        Field chi_sum(nO), bnd((size_t)(nr - 1) * 2);
        for (int m = 0; m < M; ++m)
            for (const std::string& op : ops) {
                const Case& c = L->cases[m];
                const Field &c0 = c.files.at("rchi-[" + op + "_0]-O.bin"), &cdB = c.files.at("rchi-[" + op + "_dB]-O.bin"),
                            &cB0 = c.files.at("rchi-[" + op + "_B0]-O.bin"), &rp = c.files.at("rpsi_after-[" + op + "]-O.bin");
                for (int meth = 1; meth <= 2; ++meth) {
                    if (meth == 1) {                                                                                // :736-737
                        LCHK(xee_add_fields(c0.data(), cdB.data(), tmpO.data(), nO));
                        LCHK(xee_add_fields(tmpO.data(), cB0.data(), chi_sum.data(), nO));
                    } else LCHK(xee_add_fields(cdB.data(), cB0.data(), chi_sum.data(), nO));
                    float s = 0.f;
                    if (cfg.legacy_int) LCHK(xee_exchange_conversion_legacy(rp.data(), chi_sum.data(), C + m * nO, ra.data(), za.data(), rho.data(), bnd.data(), &s, nr, nz));
                    else LCHK(xee_exchange_conversion(rp.data(), chi_sum.data(), C + m * nO, ra.data(), za.data(), rho.data(), bnd.data(), &s, nr, nz));
                    const std::string tag = meth == 1 ? "" : "2";
                    put(L, m, "bndconv" + tag + "-[" + op + "].bin", bnd.data(), nr - 1, 2);
                    put_scalar(L, m, "sum_bndconv" + tag + "_" + op, s);
                }
            }
    }
    return XEE_OK;
}

}  // namespace

extern "C" {

int xee_legacy_run(xee_legacy_t** out, const xee_legacy_config_t* cfg, int nr, int nz, int n_cases, const float* A, const float* B,
                   const float* C, const float* Q, const float* F, const float* psi_bc, const float* chi_bc)
{
    if (!out) return bad("xee_legacy_run: null out pointer");
    *out = nullptr;
    if (!cfg || !A || !B || !C || !Q || !F || nr < 4 || nz < 4 || n_cases < 1) return bad("xee_legacy_run: bad argument");
    if (cfg->op_mode < 0 || cfg->op_mode > 2) return bad("xee_legacy_run: op_mode 0 (BAROTROPIC), 1 (BAROCLINIC) or 2 (BARO_ALL)");
    xee_legacy* L = new xee_legacy();
    L->nr = nr; L->nz = nz;
    L->bt = cfg->op_mode == 0 || cfg->op_mode == 2;
    L->bcl = cfg->op_mode == 1 || cfg->op_mode == 2;
    L->has_chi_bc = chi_bc != nullptr;
    L->cases.resize(n_cases);
    xee_ensemble_t* ens = nullptr;
    int st = xee_ensemble_create(&ens, nr, nz, n_cases, cfg->device);
    if (st == XEE_OK) st = run(L, *cfg, nr, nz, n_cases, A, B, C, Q, F, psi_bc, chi_bc, ens);
    if (ens) xee_ensemble_destroy(ens);
    if (st != XEE_OK) { delete L; return st; }
    *out = L;
    return XEE_OK;
}

int xee_legacy_destroy(xee_legacy_t* L)
{
    delete L;
    return XEE_OK;
}

int xee_legacy_field_count(const xee_legacy_t* L, int* n)
{
    if (!L || !n) return bad("null argument");
    *n = (int)L->field_order.size();
    return XEE_OK;
}

int xee_legacy_field_info(const xee_legacy_t* L, int index, char* name, int name_len, int* n1, int* n2)
{
    if (!L || index < 0 || index >= (int)L->field_order.size() || !name || name_len < 2) return bad("xee_legacy_field_info: bad argument");
    const std::string& s = L->field_order[index];
    strncpy(name, s.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
    if (n1) *n1 = L->shape.at(s).first;
    if (n2) *n2 = L->shape.at(s).second;
    return XEE_OK;
}

int xee_legacy_field(const xee_legacy_t* L, int icase, const char* name, float* out)
{
    if (!L || icase < 0 || icase >= (int)L->cases.size() || !name || !out) return bad("xee_legacy_field: bad argument");
    auto it = L->cases[icase].files.find(name);
    if (it == L->cases[icase].files.end()) return bad("xee_legacy_field: no such field");
    memcpy(out, it->second.data(), it->second.size() * sizeof(float));
    return XEE_OK;
}

int xee_legacy_scalar_count(const xee_legacy_t* L, int* n)
{
    if (!L || !n) return bad("null argument");
    *n = (int)L->scalar_order.size();
    return XEE_OK;
}

int xee_legacy_scalar_info(const xee_legacy_t* L, int index, char* key, int key_len)
{
    if (!L || index < 0 || index >= (int)L->scalar_order.size() || !key || key_len < 2) return bad("xee_legacy_scalar_info: bad argument");
    strncpy(key, L->scalar_order[index].c_str(), (size_t)key_len - 1);
    key[key_len - 1] = 0;
    return XEE_OK;
}

int xee_legacy_scalar(const xee_legacy_t* L, int icase, const char* key, float* value)
{
    if (!L || icase < 0 || icase >= (int)L->cases.size() || !key || !value) return bad("xee_legacy_scalar: bad argument");
    auto it = L->cases[icase].scalars.find(key);
    if (it == L->cases[icase].scalars.end()) return bad("xee_legacy_scalar: no such scalar");
    *value = it->second;
    return XEE_OK;
}

int xee_legacy_relax(const xee_legacy_t* L, int icase, int* n_solves, int* iters, float* err_now, int capacity)
{
    if (!L || icase < 0 || icase >= (int)L->cases.size() || !n_solves) return bad("xee_legacy_relax: bad argument");
    const auto& r = L->cases[icase].relax;
    *n_solves = (int)r.size();
    for (int k = 0; k < (int)r.size() && k < capacity; ++k) {
        if (iters) iters[k] = r[k].first;
        if (err_now) err_now[k] = r[k].second;
    }
    return XEE_OK;
}

int xee_legacy_record(const xee_legacy_t* L, int icase, xee_case_record_t* rec)
{
    if (!L || icase < 0 || icase >= (int)L->cases.size() || !rec) return bad("xee_legacy_record: bad argument");
    *rec = L->cases[icase].record;
    return XEE_OK;
}

int xee_legacy_efficiency_text(const xee_legacy_t* L, int icase, double elapsed_s, char* text, int text_len)
{
    if (!L || icase < 0 || icase >= (int)L->cases.size() || !text || text_len < 2) return bad("xee_legacy_efficiency_text: bad argument");
    const std::string t = efficiency_text(L, L->cases[icase], elapsed_s);
    if ((int)t.size() + 1 > text_len) return bad("xee_legacy_efficiency_text: buffer too small");
    memcpy(text, t.c_str(), t.size() + 1);
    return XEE_OK;
}

int xee_legacy_write_outputs(const xee_legacy_t* L, int icase, const char* directory, double elapsed_s)
{
    if (!L || icase < 0 || icase >= (int)L->cases.size() || !directory) return bad("xee_legacy_write_outputs: bad argument");
    const Case& c = L->cases[icase];
    const std::string dir = directory;
    for (const auto& kv : c.files) {
        FILE* fh = fopen((dir + "/" + kv.first).c_str(), "wb");
        if (!fh) return bad(("xee_legacy_write_outputs: cannot write " + dir + "/" + kv.first).c_str());
        const size_t n = fwrite(kv.second.data(), sizeof(float), kv.second.size(), fh);
        fclose(fh);
        if (n != kv.second.size()) return bad("xee_legacy_write_outputs: short write");
    }
    FILE* fh = fopen((dir + "/efficiency.txt").c_str(), "w");
    if (!fh) return bad(("xee_legacy_write_outputs: cannot write " + dir + "/efficiency.txt").c_str());
    const std::string t = efficiency_text(L, c, elapsed_s);
    fwrite(t.data(), 1, t.size(), fh);
    fclose(fh);
    return XEE_OK;
}

}  // extern "C"
