! xee_c_api.f90 -- ISO_C_BINDING view of libxee's Tier-2 (ensemble) C-ABI and of xee_profiles
! (include/xee.h), for Fortran hosts that keep their data on the GPU between the stages of
! src/diagnose/diagnose.f90 instead of round-tripping through the Tier-1 drop-in calls.
module xee_c_api
use, intrinsic :: iso_c_binding
implicit none
public

integer(c_int), parameter :: XEE_OK = 0, XEE_ERR_INVALID_CRITERIA = 1
integer(c_int), parameter :: XEE_BAROTROPIC = 0, XEE_BAROCLINIC = 1
integer(c_int), parameter :: XEE_F_A = 0, XEE_F_B = 1, XEE_F_C = 2, XEE_F_BC_INIT = 3,          &
    &   XEE_F_FORCING = 4, XEE_F_PSI = 5, XEE_F_SOLVER_A = 6, XEE_F_SOLVER_B = 7,                &
    &   XEE_F_SOLVER_C = 8, XEE_F_ETA = 9, XEE_F_U = 10, XEE_F_W = 11, XEE_F_Q = 12
integer(c_int), parameter :: XEE_SOLVE_DEFAULT = 0, XEE_SOLVE_KEEP_PSI = 2, XEE_SOLVE_TRACE = 4

! xee_case_record_t, 128 bytes
type, bind(C) :: xee_case_record
    integer(c_int32_t) :: case_id, iters, err_bits, reserved
    real(c_float)      :: err_now, ratio, sum_Q, sum_Qeta, efficiency
    real(c_float)      :: scalars(23)
end type

! xee_legacy_config_t (include/xee.h): the legacy program's mode, domain and solver lines
type, bind(C) :: xee_legacy_config
    integer(c_int) :: geometry, density_mode
    real(c_float)  :: domain(4), testing_dt
    integer(c_int) :: op_mode
    integer(c_int) :: psi_strategy
    real(c_float)  :: psi_r
    integer(c_int) :: psi_max_iter
    real(c_float)  :: psi_alpha
    integer(c_int) :: chi_strategy
    real(c_float)  :: chi_r
    integer(c_int) :: chi_max_iter
    real(c_float)  :: chi_alpha
    integer(c_int) :: legacy_int, chained, device, first_case_id
end type

interface
    function xee_device_count(n) bind(C, name="xee_device_count") result(st)
        import :: c_int
        integer(c_int), intent(out) :: n
        integer(c_int) :: st
    end function

    function xee_last_error() bind(C, name="xee_last_error") result(p)
        import :: c_ptr
        type(c_ptr) :: p
    end function

    function xee_profiles(geometry, density_mode, domain, nr, nz, ra, za, rcuva, exner, rho,     &
        &                 sin_table, dr, dz) bind(C, name="xee_profiles") result(st)
        import :: c_int, c_float
        integer(c_int), value :: geometry, density_mode, nr, nz
        real(c_float), intent(in)  :: domain(*)
        real(c_float), intent(out) :: ra(*), za(*), rcuva(*), exner(*), rho(*), sin_table(*)
        real(c_float), intent(out) :: dr, dz
        integer(c_int) :: st
    end function

    function xee_ensemble_create(ens, nr, nz, n_members, device) bind(C, name="xee_ensemble_create") result(st)
        import :: c_int, c_ptr
        type(c_ptr), intent(out) :: ens
        integer(c_int), value :: nr, nz, n_members, device
        integer(c_int) :: st
    end function

    function xee_ensemble_destroy(ens) bind(C, name="xee_ensemble_destroy") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int) :: st
    end function

    function xee_ensemble_set_profiles(ens, ra, za, rcuva, rho, exner, dr, dz)                   &
        &                              bind(C, name="xee_ensemble_set_profiles") result(st)
        import :: c_int, c_ptr, c_float
        type(c_ptr), value :: ens
        real(c_float), intent(in) :: ra(*), za(*), rcuva(*), rho(*), exner(*)
        real(c_float), value :: dr, dz
        integer(c_int) :: st
    end function

    function xee_ensemble_upload(ens, field, first, count, host) bind(C, name="xee_ensemble_upload") result(st)
        import :: c_int, c_ptr, c_float
        type(c_ptr), value :: ens
        integer(c_int), value :: field, first, count
        real(c_float), intent(in) :: host(*)
        integer(c_int) :: st
    end function

    function xee_ensemble_download(ens, field, first, count, host) bind(C, name="xee_ensemble_download") result(st)
        import :: c_int, c_ptr, c_float
        type(c_ptr), value :: ens
        integer(c_int), value :: field, first, count
        real(c_float), intent(inout) :: host(*)
        integer(c_int) :: st
    end function

    function xee_ensemble_forcing_from_B(ens) bind(C, name="xee_ensemble_forcing_from_B") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int) :: st
    end function

    function xee_ensemble_assemble(ens, variant) bind(C, name="xee_ensemble_assemble") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int), value :: variant
        integer(c_int) :: st
    end function

    function xee_ensemble_solve(ens, max_iter, check_step, converge_time, lost_rate, r1, r2,     &
        &                       alpha, flags) bind(C, name="xee_ensemble_solve") result(st)
        import :: c_int, c_ptr, c_float
        type(c_ptr), value :: ens
        integer(c_int), value :: max_iter, check_step, converge_time, lost_rate, flags
        real(c_float), value :: r1, r2, alpha
        integer(c_int) :: st
    end function

    function xee_ensemble_cal_eta(ens) bind(C, name="xee_ensemble_cal_eta") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int) :: st
    end function

    function xee_ensemble_cal_uw(ens) bind(C, name="xee_ensemble_cal_uw") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int) :: st
    end function

    function xee_ensemble_ee_scalars(ens) bind(C, name="xee_ensemble_ee_scalars") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int) :: st
    end function

    function xee_ensemble_records(ens, rec) bind(C, name="xee_ensemble_records") result(st)
        import :: c_int, c_ptr, xee_case_record
        type(c_ptr), value :: ens
        type(xee_case_record), intent(out) :: rec(*)
        integer(c_int) :: st
    end function

    ! host-array stencils of the two driver programs (quick-tools2.f90:1-85; old-diagnose/diagnose.f90:412-436,
    ! :1143-1174); which = 0..5 in the order d_dz_B2A, d_dz_O2C, d_dr_B2B, d_dr_B2C, d_dr_O2A, d_rcuvdr_O2A
    function xee_ensemble_set_case_ids(ens, first_case_id) bind(C, name="xee_ensemble_set_case_ids") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int), value :: first_case_id
        integer(c_int) :: st
    end function

    ! ---- the legacy program's stages (src/old-diagnose/diagnose.f90), host arrays in the reference's layouts
    function xee_ensemble_assemble_solver(ens, use_b) bind(C, name="xee_ensemble_assemble_solver") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: ens
        integer(c_int), value :: use_b
        integer(c_int) :: st
    end function

    function xee_legacy_averages(rhoA_in, rhoB_in, rhoC_in, rhoA_A, rhoB_C, rhoB_B, rhoC_C, nr, nz) &
        &                        bind(C, name="xee_legacy_averages") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: rhoA_in(*), rhoB_in(*), rhoC_in(*)
        real(c_float), intent(out) :: rhoA_A(*), rhoB_C(*), rhoB_B(*), rhoC_C(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_m2_from_C(rhoC_C, ra, rcuva, sin_table, m2_B, nr, nz) bind(C, name="xee_m2_from_C") result(st)
        import :: c_int, c_float, c_ptr
        real(c_float), intent(in)  :: rhoC_C(*), ra(*), rcuva(*)
        type(c_ptr), value :: sin_table                     ! c_null_ptr for the cylindrical geometry
        real(c_float), intent(out) :: m2_B(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_JJ_B(Q_B, exner, JJ_B, nr, nz) bind(C, name="xee_JJ_B") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: Q_B(*), exner(*)
        real(c_float), intent(out) :: JJ_B(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_rhs_thermal(Q_B, ra, exner, rhs_O, nr, nz) bind(C, name="xee_rhs_thermal") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: Q_B(*), ra(*), exner(*)
        real(c_float), intent(out) :: rhs_O(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_rhs_from_B(b_B, f_O, nr, nz) bind(C, name="xee_rhs_from_B") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: b_B(*)
        real(c_float), intent(out) :: f_O(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_add_fields(a, b, out, n) bind(C, name="xee_add_fields") result(st)
        import :: c_int, c_float, c_size_t
        real(c_float), intent(in)  :: a(*), b(*)
        real(c_float), intent(out) :: out(*)
        integer(c_size_t), value :: n
        integer(c_int) :: st
    end function

    function xee_integrate_weight_B(weight, ra, za, rcuva, rho, nr, nz, total) bind(C, name="xee_integrate_weight_B") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: weight(*), ra(*), za(*), rcuva(*), rho(*)
        integer(c_int), value :: nr, nz
        real(c_float), intent(out) :: total
        integer(c_int) :: st
    end function

    function xee_cal_wtheta(w_A, theta_B, wtheta_B, nr, nz) bind(C, name="xee_cal_wtheta") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: w_A(*), theta_B(*)
        real(c_float), intent(out) :: wtheta_B(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_exchange_conversion_legacy(rpsi, rchi, rhoC, ra, za, rho, bndconv, total, nr, nz)   &
        &                                   bind(C, name="xee_exchange_conversion_legacy") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)  :: rpsi(*), rchi(*), rhoC(*), ra(*), za(*), rho(*)
        real(c_float), intent(out) :: bndconv(*), total
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    ! ---- multi-GPU: one process per GPU, the record gather over RCCL (include/xee.h, xee_comm_* / xee_gather_scalars)
    function xee_comm_unique_id(id) bind(C, name="xee_comm_unique_id") result(st)
        import :: c_int, c_char
        character(kind=c_char), intent(out) :: id(*)          ! XEE_COMM_ID_BYTES = 128
        integer(c_int) :: st
    end function

    function xee_comm_create(comm, id, rank, world, device) bind(C, name="xee_comm_create") result(st)
        import :: c_int, c_ptr, c_char
        type(c_ptr), intent(out) :: comm
        character(kind=c_char), intent(in) :: id(*)
        integer(c_int), value :: rank, world, device
        integer(c_int) :: st
    end function

    function xee_comm_set_timeout(seconds) bind(C, name="xee_comm_set_timeout") result(st)
        import :: c_int, c_double
        real(c_double), value :: seconds
        integer(c_int) :: st
    end function

    function xee_comm_destroy(comm) bind(C, name="xee_comm_destroy") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: comm
        integer(c_int) :: st
    end function

    function xee_gather_scalars(comm, local, per_rank, all) bind(C, name="xee_gather_scalars") result(st)
        import :: c_int, c_ptr, xee_case_record
        type(c_ptr), value :: comm
        type(xee_case_record), intent(in) :: local(*)
        integer(c_int), value :: per_rank
        type(xee_case_record), intent(out) :: all(*)
        integer(c_int) :: st
    end function

    function xee_fd_stencil(which, from_dat, to_dat, ra, za, rcuva, nr, nz) bind(C, name="xee_fd_stencil") result(st)
        import :: c_int, c_float
        integer(c_int), value :: which, nr, nz
        real(c_float), intent(in) :: from_dat(*), ra(*), za(*), rcuva(*)
        real(c_float), intent(inout) :: to_dat(*)
        integer(c_int) :: st
    end function

    function xee_rhs_momentum(m2_B, F_B, za, rcuva, rhs_O, nr, nz) bind(C, name="xee_rhs_momentum") result(st)
        import :: c_int, c_float
        real(c_float), intent(in) :: m2_B(*), F_B(*), za(*), rcuva(*)
        real(c_float), intent(out) :: rhs_O(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_exchange_conversion(rpsi, rchi, rhoC, ra, za, rho, bndconv, total, nr, nz)   &
        &                            bind(C, name="xee_exchange_conversion") result(st)
        import :: c_int, c_float
        real(c_float), intent(in) :: rpsi(*), rchi(*), rhoC(*), ra(*), za(*), rho(*)
        real(c_float), intent(out) :: bndconv(*), total
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    ! Stage II of the legacy TENDENCY mode (old-diagnose/diagnose.f90:462-518, :893-912)
    function xee_theta_tendency(JJ_B, rhoA_A, w_A, rhoB_C, u_C, dtheta_dt_B, nr, nz)   &
        &                       bind(C, name="xee_theta_tendency") result(st)
        import :: c_int, c_float
        real(c_float), intent(in) :: JJ_B(*), rhoA_A(*), w_A(*), rhoB_C(*), u_C(*)
        real(c_float), intent(out) :: dtheta_dt_B(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_relative_theta(dtheta_dz_A, dtheta_dr_C, ra, za, theta_B, nr, nz)   &
        &                       bind(C, name="xee_relative_theta") result(st)
        import :: c_int, c_float
        real(c_float), intent(in) :: dtheta_dz_A(*), dtheta_dr_C(*), ra(*), za(*)
        real(c_float), intent(out) :: theta_B(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function

    function xee_theta_anomaly_update(theta_B, testing_dt, rhoB_B, rhoA_A, rhoB_C, b_anomaly_B, solver_b_anomaly_B,   &
        &                             ra, za, rcuva, rho, nr, nz) bind(C, name="xee_theta_anomaly_update") result(st)
        import :: c_int, c_float
        real(c_float), intent(inout) :: theta_B(*), rhoB_B(*), rhoA_A(*), rhoB_C(*)
        real(c_float), value :: testing_dt
        real(c_float), intent(out) :: b_anomaly_B(*), solver_b_anomaly_B(*)
        real(c_float), intent(in) :: ra(*), za(*), rcuva(*), rho(*)
        integer(c_int), value :: nr, nz
        integer(c_int) :: st
    end function
    ! frees the device ensemble the Tier-1 calls of this thread keep between calls (include/xee.h)
    function xee_legacy_run(res, cfg, nr, nz, n_cases, A, B, C, Q, F, psi_bc, chi_bc) bind(C, name="xee_legacy_run") result(st)
        import :: c_int, c_ptr, c_float, xee_legacy_config
        type(c_ptr), intent(out) :: res
        type(xee_legacy_config), intent(in) :: cfg
        integer(c_int), value :: nr, nz, n_cases
        real(c_float), intent(in) :: A(*), B(*), C(*), Q(*), F(*)
        type(c_ptr), value :: psi_bc, chi_bc              ! (nr,nz) planes per case, or c_null_ptr
        integer(c_int) :: st
    end function

    function xee_legacy_destroy(res) bind(C, name="xee_legacy_destroy") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: res
        integer(c_int) :: st
    end function

    function xee_legacy_relax(res, icase, n_solves, iters, err_now, capacity) bind(C, name="xee_legacy_relax") result(st)
        import :: c_int, c_ptr
        type(c_ptr), value :: res, iters, err_now         ! (this form: query the count with null arrays)
        integer(c_int), value :: icase, capacity
        integer(c_int), intent(out) :: n_solves
        integer(c_int) :: st
    end function

    function xee_legacy_relax_get(res, icase, n_solves, iters, err_now, capacity) bind(C, name="xee_legacy_relax") result(st)
        import :: c_int, c_ptr, c_float
        type(c_ptr), value :: res
        integer(c_int), value :: icase, capacity
        integer(c_int), intent(out) :: n_solves
        integer(c_int), intent(out) :: iters(*)
        real(c_float), intent(out) :: err_now(*)
        integer(c_int) :: st
    end function

    function xee_legacy_write_outputs(res, icase, directory, elapsed_s) bind(C, name="xee_legacy_write_outputs") result(st)
        import :: c_int, c_ptr, c_char, c_double
        type(c_ptr), value :: res
        integer(c_int), value :: icase
        character(kind=c_char), intent(in) :: directory(*)
        real(c_double), value :: elapsed_s
        integer(c_int) :: st
    end function

    function xee_tier1_release() bind(C, name="xee_tier1_release") result(st)
        import :: c_int
        integer(c_int) :: st
    end function
end interface

contains

! abort with libxee's message; there is no CPU fallback to fall back to
subroutine xee_check(st, where)
implicit none
integer(c_int), intent(in) :: st
character(*), intent(in) :: where
character(kind=c_char), pointer :: msg(:)
type(c_ptr) :: p
integer :: n

if (st == XEE_OK) return
p = xee_last_error()
n = 0
if (c_associated(p)) then
    call c_f_pointer(p, msg, [512])
    do while (n < 512)
        if (msg(n+1) == c_null_char) exit
        n = n + 1
    end do
    write (*, '(A,A,A,I0,A,512A1)') "ERROR: libxee ", where, " failed with status ", st, ": ", msg(1:n)
else
    write (*, '(A,A,A,I0)') "ERROR: libxee ", where, " failed with status ", st
end if
error stop 1
end subroutine

end module xee_c_api
