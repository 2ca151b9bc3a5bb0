! xee_old_diagnose.f90 -- Fortran 2008 host of the LEGACY program on the MI355X: `xee-old-diagnose < diag.txt`.
!
! Keeps the surface of src/old-diagnose/diagnose.f90: the positional stdin config with // comments (:76-167, SURVEY.md
! Appendix A.2), the raw float32 input files, every output file name and shape (:389-770) and efficiency.txt with all its
! labels (:779-841, what xtt-lib-python/XEffReader.py parses).  Every numerical stage is a libxee call (include/xee.h):
! the solves run on a device-resident one-member ensemble (the first guess is carried from solve to solve exactly as the
! program carries it, :545), the O(N) stencils through the Tier-1 entries.  No CPU fallback.
!
!     source terms   sum_Q :284, solver coefficients :289-312, staggered averages :329-354, m2 :359-380 (INTENDED form),
!                    J :383-387, RHS_rpsi_thm :395-406, RHS_rpsi_mom :412-436
!     Stage I        psi for thm + mom on the full-B operator :449-463
!     Stage II       u, w :468, dtheta/dt :470-479, anomaly dB / updated rho fields / theta :489-518
!     Stage III      f_basic, f_anomaly :523-541; chi solves on {L(B=0), L(B0+dB)} x {0 with boundary, -dB, -B0} :548-673
!     integral check :679-725;  exchange conversion, methods 1 and 2 :730-772;  efficiency.txt :779-841
!
! Solver lines: `strategy r max_iter alpha` with strategy 1 (stop when err_now < r at a check) or 2 (|ratio| < r, ten
! passes); they are the r1- and r2-only cases of the current solve_elliptic's state machine (converge_time 1 resp. 10).
! Strategies 3/4 (max norm over the whole array) and the INSTANT mode (reads an unassigned b_anomaly_B) are not offered.
!
! Where the legacy program has undefined behaviour this host is DEFINED (SURVEY.md quirk 14): Q and F files are (nr-1,nz-1)
! B-cell records; m2 in its intended form; the "w/ boundary" sums are 0 without a chi boundary file.  ONE optional trailing
! config line, the word `legacy_records`, makes it behave as the legacy binary does on its own inputs: Q and F are then
! nr x nz records of which the first (nr-1)(nz-1) floats are taken (what the program really holds after :241-242), and the
! exchange conversion keeps the program's INTEGER r, dr, dz (:1146) -- with it the files are byte-identical to the legacy
! binary's (tests/test_gpu_host_programs.py).  Without it the exchange conversion uses real metric factors.
program xee_old_diagnose
use, intrinsic :: iso_c_binding
use xee_c_api
use field_tools
use read_input_tools
use message_tools
implicit none

integer, parameter :: stdin = 5
real(4), parameter :: g0 = 9.8, theta0 = 298.0                       ! xtt-lib-fortran/constants.f90:4
character(256) :: mode_str, word(4), buffer, input_folder, output_folder, A_file, B_file, C_file, Q_file, F_file, &
&                 rpsi_bc_file, rchi_bc_file, yes_or_no
integer :: nr, nz, geometry, density_mode, op_mode, i, k, nops, ndev
integer :: strategy_rpsi, strategy_rchi, max_iter_rpsi, max_iter_rchi
real(4) :: testing_dt, domain(4), Lr(2), Lz(2), planet_radius, r_rpsi, r_rchi, alpha_rpsi, alpha_rchi, dr, dz
logical :: use_rpsi_bc, use_rchi_bc, legacy_records, bt, bcl
real(4), allocatable, target :: sin_table(:)
real(4), allocatable :: ra(:), za(:), rcuva(:), exner(:), rho(:)
real(4), allocatable :: rhoA_in(:,:), rhoB_in(:,:), rhoC_in(:,:), record(:), Q_in(:,:), F_in(:,:), rpsi_bc(:,:), rchi_bc(:,:)
real(4), allocatable :: rhoA_A(:,:), rhoB_C(:,:), rhoB_B(:,:), rhoC_C(:,:), b_basic_B(:,:), b_anomaly_B(:,:), m2(:,:), JJ_B(:,:)
real(4), allocatable :: solverA(:,:), solverB(:,:), solverC(:,:), sb_anomaly(:,:), sb_B0dB(:,:)
real(4), allocatable :: thm(:,:), mom(:,:), f_psi(:,:), f_basic(:,:), f_anomaly(:,:), f(:,:), zero_O(:,:)
real(4), allocatable :: rpsi(:,:), rchi(:,:), chi_sum(:,:), tmp_O(:,:), w_A(:,:), u_C(:,:), eta(:,:), theta(:,:), wtheta(:,:), bnd(:,:)
real(4), allocatable :: rpsi_after(:,:,:), rchi_keep(:,:,:,:)
type(c_ptr) :: ens
type(xee_case_record) :: rec(1)
real(4) :: sum_Q, sum_dtheta_dt, sum_Qeta(2,3), sum_wtheta(2), sum_bnd(2), sum_bnd2(2), tmp1
character(8) :: opname(2)
character(4) :: rhsname(3)
integer(8) :: clk0, clk1, clk_rate

call system_clock(clk0, clk_rate)
! ---- config (:76-167) ------------------------------------------------------------------------------------------
call read_input(stdin, mode_str)
do i = 1, size(word)
    if (split_line(mode_str, word(i), "-") /= 0 .and. i /= size(word)) then
        call error_msg("INIT", 1, "Mode number is not correct"); stop
    end if
end do
select case (trim(word(1)))
case ('CYLINDRICAL'); geometry = 0
case ('SPHERICAL');   geometry = 1
case default
    call error_msg("INIT", 1, "Unknown Mode : [" // trim(word(1)) // "]"); stop
end select
if (trim(word(2)) /= 'TENDENCY') then
    call error_msg("INIT", 1, "only the TENDENCY mode is offered (INSTANT reads an unassigned anomaly field), not [" // trim(word(2)) // "]"); stop
end if
select case (trim(word(3)))
case ('DENSITY_NORMAL');     density_mode = 0
case ('DENSITY_BOUSSINESQ'); density_mode = 1
case default
    call error_msg("INIT", 1, "Unknown Mode : [" // trim(word(3)) // "]"); stop
end select
select case (trim(word(4)))
case ('BARO_ALL');   op_mode = 2
case ('BAROCLINIC'); op_mode = 1
case ('BAROTROPIC'); op_mode = 0
case default
    call error_msg("INIT", 1, "Unknown Mode : [" // trim(word(4)) // "]"); stop
end select
call read_input(stdin, buffer); read (buffer, *) testing_dt
domain = 0.0
call read_input(stdin, buffer)
if (geometry == 0) then
    read (buffer, *) Lr(1), Lr(2), Lz(1), Lz(2)
    domain = [Lr(1), Lr(2), Lz(1), Lz(2)]
else
    read (buffer, *) planet_radius, Lz(1), Lz(2)
    domain(1:3) = [planet_radius, Lz(1), Lz(2)]
end if
call read_input(stdin, buffer); read (buffer, *) nr, nz
call read_input(stdin, input_folder)
call read_input(stdin, output_folder)
call read_input(stdin, A_file); call read_input(stdin, B_file); call read_input(stdin, C_file)
call read_input(stdin, Q_file); call read_input(stdin, F_file)
call read_input(stdin, buffer); read (buffer, *) strategy_rpsi, r_rpsi, max_iter_rpsi, alpha_rpsi
call read_input(stdin, buffer); read (buffer, *) strategy_rchi, r_rchi, max_iter_rchi, alpha_rchi
call read_input(stdin, yes_or_no)
use_rpsi_bc = trim(yes_or_no) == 'yes'
if (use_rpsi_bc) call read_input(stdin, rpsi_bc_file)
call read_input(stdin, yes_or_no)
use_rchi_bc = trim(yes_or_no) == 'yes'
if (use_rchi_bc) call read_input(stdin, rchi_bc_file)
legacy_records = try_read_input(stdin, buffer)
if (legacy_records) legacy_records = trim(adjustl(buffer)) == 'legacy_records'
do i = 1, 2
    if (merge(strategy_rpsi, strategy_rchi, i == 1) /= 1 .and. merge(strategy_rpsi, strategy_rchi, i == 1) /= 2) then
        call error_msg("INIT", 1, "only strategies 1 and 2 are offered"); stop
    end if
    if (mod(merge(max_iter_rpsi, max_iter_rchi, i == 1), 100) /= 0) then
        call error_msg("INIT", 1, "max_iter must be a multiple of 100 (the legacy loop looks at it on check iterations only)"); stop
    end if
end do
print *, "mode: ", geometry, ",", 0, ",", density_mode, ",", op_mode
print *, "Testing time: ", testing_dt
print *, "nr:", nr, ", nz:", nz
print *, "Input folder: ", trim(input_folder)
print *, "Output folder: ", trim(output_folder)
print *, "rpsi's strategy, residue, iter: ", strategy_rpsi, r_rpsi, max_iter_rpsi, alpha_rpsi
print *, "rchi's strategy, residue, iter: ", strategy_rchi, r_rchi, max_iter_rchi, alpha_rchi

call xee_check(xee_device_count(ndev), "device_count")
if (ndev < 1) then
    call error_msg("INIT", 1, "no MI355X visible: this program has no CPU fallback"); error stop 1
end if

allocate(ra(nr), za(nz), rcuva(nr), exner(nz), rho(nz), sin_table(nr))
allocate(rhoA_in(nr,nz), rhoB_in(nr,nz), rhoC_in(nr,nz), Q_in(nr-1,nz-1), F_in(nr-1,nz-1))
allocate(rhoA_A(nr-1,nz), rhoB_C(nr,nz-1), rhoB_B(nr-1,nz-1), rhoC_C(nr,nz-1), b_basic_B(nr-1,nz-1), b_anomaly_B(nr-1,nz-1))
allocate(m2(nr-1,nz-1), JJ_B(nr-1,nz-1), solverA(nr-1,nz-2), solverB(nr-1,nz-1), solverC(nr-2,nz-1), sb_anomaly(nr-1,nz-1), sb_B0dB(nr-1,nz-1))
allocate(thm(nr,nz), mom(nr,nz), f_psi(nr,nz), f_basic(nr,nz), f_anomaly(nr,nz), f(nr,nz), zero_O(nr,nz), rpsi(nr,nz), rchi(nr,nz))
allocate(chi_sum(nr,nz), tmp_O(nr,nz), w_A(nr-1,nz), u_C(nr,nz-1), eta(nr-1,nz), theta(nr-1,nz-1), wtheta(nr-1,nz-1), bnd(nr-1,2))
allocate(rpsi_after(nr,nz,2), rchi_keep(nr,nz,2,3))
zero_O = 0.0

call read_2Dfield(15, trim(input_folder)//"/"//A_file, rhoA_in, nr, nz)
call read_2Dfield(15, trim(input_folder)//"/"//B_file, rhoB_in, nr, nz)
call read_2Dfield(15, trim(input_folder)//"/"//C_file, rhoC_in, nr, nz)
if (legacy_records) then          ! what the legacy program holds after reading nr x nz records into (nr-1,nz-1) arrays
    allocate(record(nr*nz))
    call read_2Dfield(15, trim(input_folder)//"/"//Q_file, record, nr, nz)
    Q_in = reshape(record(1:(nr-1)*(nz-1)), [nr-1, nz-1])
    call read_2Dfield(15, trim(input_folder)//"/"//F_file, record, nr, nz)
    F_in = reshape(record(1:(nr-1)*(nz-1)), [nr-1, nz-1])
else
    call read_2Dfield(15, trim(input_folder)//"/"//Q_file, Q_in, nr-1, nz-1)
    call read_2Dfield(15, trim(input_folder)//"/"//F_file, F_in, nr-1, nz-1)
end if
if (use_rpsi_bc) then
    allocate(rpsi_bc(nr,nz)); call read_2Dfield(15, trim(input_folder)//"/"//rpsi_bc_file, rpsi_bc, nr, nz)
end if
if (use_rchi_bc) then
    allocate(rchi_bc(nr,nz)); call read_2Dfield(15, trim(input_folder)//"/"//rchi_bc_file, rchi_bc, nr, nz)
end if

! ---- profiles, ensemble, source terms ------------------------------------------------------------------------------
call xee_check(xee_profiles(geometry, density_mode, domain, nr, nz, ra, za, rcuva, exner, rho, sin_table, dr, dz), "profiles")
call xee_check(xee_ensemble_create(ens, nr, nz, 1, -1), "ensemble_create")
call xee_check(xee_ensemble_set_profiles(ens, ra, za, rcuva, rho, exner, dr, dz), "set_profiles")
call xee_check(xee_ensemble_upload(ens, XEE_F_A, 0, 1, rhoA_in), "upload A")
call xee_check(xee_ensemble_upload(ens, XEE_F_B, 0, 1, rhoB_in), "upload B")
call xee_check(xee_ensemble_upload(ens, XEE_F_C, 0, 1, rhoC_in), "upload C")
call xee_check(xee_ensemble_upload(ens, XEE_F_Q, 0, 1, Q_in), "upload Q")
call xee_check(xee_integrate_weight_B(Q_in, ra, za, rcuva, rho, nr, nz, sum_Q), "sum_Q")                      ! :284
call xee_check(xee_ensemble_assemble(ens, XEE_BAROCLINIC), "assemble")                                         ! :289-312, :449
call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_A, 0, 1, solverA), "download solver A")
call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_B, 0, 1, solverB), "download solver B")
call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_C, 0, 1, solverC), "download solver C")
call xee_check(xee_legacy_averages(rhoA_in, rhoB_in, rhoC_in, rhoA_A, rhoB_C, rhoB_B, rhoC_C, nr, nz), "averages")   ! :329-354
b_basic_B = rhoB_B                                                                                             ! :348
if (geometry == 1) then
    call xee_check(xee_m2_from_C(rhoC_C, ra, rcuva, c_loc(sin_table), m2, nr, nz), "m2")                       ! :359-380
else
    call xee_check(xee_m2_from_C(rhoC_C, ra, rcuva, c_null_ptr, m2, nr, nz), "m2")
end if
call xee_check(xee_JJ_B(Q_in, exner, JJ_B, nr, nz), "JJ_B")                                                    ! :383-387
call write_2Dfield(11, trim(output_folder)//"/J-B.bin", JJ_B, nr-1, nz-1)
call write_2Dfield(11, trim(output_folder)//"/solver_a-sA.bin", solverA, nr-1, nz-2)
call write_2Dfield(11, trim(output_folder)//"/solver_b-B.bin", solverB, nr-1, nz-1)
call write_2Dfield(11, trim(output_folder)//"/solver_c-sC.bin", solverC, nr-2, nz-1)
call xee_check(xee_rhs_thermal(Q_in, ra, exner, thm, nr, nz), "rhs_thermal")                                   ! :395-406
call write_2Dfield(11, trim(output_folder)//"/RHS_rpsi_thm-O.bin", thm, nr, nz)
call xee_check(xee_rhs_momentum(m2, F_in, za, rcuva, mom, nr, nz), "rhs_momentum")                             ! :412-436
call write_2Dfield(11, trim(output_folder)//"/RHS_rpsi_mom-O.bin", mom, nr, nz)
call xee_check(xee_add_fields(thm, mom, f_psi, int(nr, c_size_t) * nz), "f_psi")                               ! :456
print *, "Initialization complete."

! ---- Stage I (:449-463) ----------------------------------------------------------------------------------------------
print *, "Solving rpsi..."
call xee_check(xee_ensemble_upload(ens, XEE_F_FORCING, 0, 1, f_psi), "upload f")
if (use_rpsi_bc) then
    call solve(strategy_rpsi, r_rpsi, max_iter_rpsi, alpha_rpsi, rpsi_bc, .true., rpsi)
else
    call solve(strategy_rpsi, r_rpsi, max_iter_rpsi, alpha_rpsi, zero_O, .true., rpsi)
end if
call write_2Dfield(11, trim(output_folder)//"/rpsi_before-O.bin", rpsi, nr, nz)
! ---- Stage II (:468-518) ---------------------------------------------------------------------------------------------
call xee_check(xee_ensemble_cal_uw(ens), "cal_uw")                                                             ! rpsiToUW :915-941
call xee_check(xee_ensemble_download(ens, XEE_F_W, 0, 1, w_A), "download w")
call xee_check(xee_ensemble_download(ens, XEE_F_U, 0, 1, u_C), "download u")
call xee_check(xee_theta_tendency(JJ_B, rhoA_A, w_A, rhoB_C, u_C, theta, nr, nz), "theta_tendency")            ! :470-479
print *, "Max dtheta_dt: ", maxval(theta)
call write_2Dfield(11, trim(output_folder)//"/w_before-A.bin", w_A, nr-1, nz)
call write_2Dfield(11, trim(output_folder)//"/u_before-C.bin", u_C, nr, nz-1)
call write_2Dfield(11, trim(output_folder)//"/dtheta_dt-B.bin", theta, nr-1, nz-1)
call xee_check(xee_integrate_weight_B(theta, ra, za, rcuva, rho, nr, nz, sum_dtheta_dt), "sum dtheta_dt")     ! :487
call xee_check(xee_theta_anomaly_update(theta, testing_dt, rhoB_B, rhoA_A, rhoB_C, b_anomaly_B, sb_anomaly, &
&                                       ra, za, rcuva, rho, nr, nz), "anomaly_update")                        ! :489-518
call write_2Dfield(11, trim(output_folder)//"/theta_after-B.bin", theta, nr-1, nz-1)
! ---- Stage III right-hand sides (:523-541) ---------------------------------------------------------------------------
call xee_check(xee_rhs_from_B(b_basic_B, f_basic, nr, nz), "f_basic")
call xee_check(xee_rhs_from_B(b_anomaly_B, f_anomaly, nr, nz), "f_anomaly")
call xee_check(xee_add_fields(f_basic, f_anomaly, f, int(nr, c_size_t) * nz), "RHS_rchi")
call write_2Dfield(11, trim(output_folder)//"/RHS_rchi-O.bin", f, nr, nz)
call xee_check(xee_add_fields(solverB, sb_anomaly, sb_B0dB, int(nr - 1, c_size_t) * (nz - 1)), "solver_b B0+dB")   ! :572

bt = op_mode == 0 .or. op_mode == 2
bcl = op_mode == 1 .or. op_mode == 2
nops = 0
if (bt) then
    nops = nops + 1; opname(nops) = "0"
end if
if (bcl) then
    nops = nops + 1; opname(nops) = "B0dB"
end if
rhsname = ["0   ", "dB  ", "B0  "]
sum_Qeta = 0.0; sum_wtheta = 0.0; sum_bnd = 0.0; sum_bnd2 = 0.0

if (use_rchi_bc) then                                                                                          ! :548-588
    do k = 1, nops
        print *, "Solving CHI with L(", trim(opname(k)), ") = 0 with boundary condition"
        call chi_solve(k, 1, zero_O, rchi_bc, k == 1)
    end do
end if
do i = 2, 3                                                                                                    ! :590-673
    do k = 1, nops
        print *, "Solving CHI with L(", trim(opname(k)), ") = -", trim(rhsname(i))
        if (i == 2) then
            call chi_solve(k, i, f_anomaly, zero_O, k == 1)           ! rchi = 0.0 (:590) before the first of them
        else
            call chi_solve(k, i, f_basic, zero_O, .false.)
        end if
    end do
end do
! ---- integral check (:679-725) -----------------------------------------------------------------------------------------
print *, "Integral check..."
do k = 1, nops
    call set_operator(k)
    call xee_check(xee_ensemble_upload(ens, XEE_F_FORCING, 0, 1, f_psi), "upload f")
    if (use_rpsi_bc) then
        call solve(strategy_rpsi, r_rpsi, max_iter_rpsi, alpha_rpsi, rpsi_bc, k == 1, rpsi)
    else
        call solve(strategy_rpsi, r_rpsi, max_iter_rpsi, alpha_rpsi, zero_O, k == 1, rpsi)
    end if
    rpsi_after(:,:,k) = rpsi
    call xee_check(xee_ensemble_cal_uw(ens), "cal_uw")
    call xee_check(xee_ensemble_download(ens, XEE_F_W, 0, 1, w_A), "download w")
    call xee_check(xee_ensemble_download(ens, XEE_F_U, 0, 1, u_C), "download u")
    call write_2Dfield(11, trim(output_folder)//"/rpsi_after-["//trim(opname(k))//"]-O.bin", rpsi, nr, nz)
    call write_2Dfield(11, trim(output_folder)//"/w_after-["//trim(opname(k))//"]-A.bin", w_A, nr-1, nz)
    call write_2Dfield(11, trim(output_folder)//"/u_after-["//trim(opname(k))//"]-C.bin", u_C, nr, nz-1)
    call xee_check(xee_cal_wtheta(w_A, theta, wtheta, nr, nz), "cal_wtheta")                                   ! :1117-1127
    call xee_check(xee_integrate_weight_B(wtheta, ra, za, rcuva, rho, nr, nz, tmp1), "sum wtheta")             ! :1094-1113
    sum_wtheta(k) = tmp1 * (g0 / theta0)                                                                       ! :703
    call write_2Dfield(11, trim(output_folder)//"/wtheta_JF_after-["//trim(opname(k))//"]-B.bin", wtheta, nr-1, nz-1)
end do
! ---- exchange conversion, methods 1 and 2 (:730-772) -------------------------------------------------------------------
if (use_rchi_bc) then
    print *, "Exchange conversion term check..."
    do k = 1, nops
        call xee_check(xee_add_fields(rchi_keep(:,:,k,1), rchi_keep(:,:,k,2), tmp_O, int(nr, c_size_t) * nz), "chi sum")
        call xee_check(xee_add_fields(tmp_O, rchi_keep(:,:,k,3), chi_sum, int(nr, c_size_t) * nz), "chi sum")
        call exchange(rpsi_after(:,:,k), chi_sum, sum_bnd(k))
        call write_2Dfield(11, trim(output_folder)//"/bndconv-["//trim(opname(k))//"].bin", bnd, nr-1, 2)
        call xee_check(xee_add_fields(rchi_keep(:,:,k,2), rchi_keep(:,:,k,3), chi_sum, int(nr, c_size_t) * nz), "chi sum")
        call exchange(rpsi_after(:,:,k), chi_sum, sum_bnd2(k))
        call write_2Dfield(11, trim(output_folder)//"/bndconv2-["//trim(opname(k))//"].bin", bnd, nr-1, 2)
    end do
end if
call xee_check(xee_ensemble_destroy(ens), "ensemble_destroy")
call system_clock(clk1)

! ---- efficiency.txt (:779-841); ES16.9 so that an fp32 value survives the round trip -----------------------------------
open (unit=15, file=trim(output_folder)//"/efficiency.txt", action="write", status="replace")
write (15,'(A,ES16.9)') " Time elapsed (sec)                          : ", real(clk1 - clk0) / real(clk_rate)
write (15,'(A,ES16.9)') " sum Q                                       : ", sum_Q
write (15,'(A,ES16.9)') " sum dtheta_dt                               : ", sum_dtheta_dt
write (15,'(A,ES16.9)') " Local heat response (sum Q / sum dtheta_dt) : ", sum_dtheta_dt / sum_Q
do k = 1, nops
    write (15,'(A)') " # Boundary efficiency"
    if (use_rchi_bc) call two(lab("eta [L(", k, " = 0]      w/  boundary"), sum_Qeta(k,1))
    write (15,'(A)') " # Internal efficiency"
    call two(lab("eta [L(", k, " = dB]     wo/ boundary"), sum_Qeta(k,2))
    call two(lab("eta [L(", k, " = B0]     wo/ boundary"), sum_Qeta(k,3))
    if (use_rchi_bc) then
        write (15,'(A)') " # Boundary conversion (Method 1)"
        if (trim(opname(k)) == "0") then
            call two("bndconv [L(B=0) = B0dB]   w/ boundary", sum_bnd(k))
        else
            call two("bndconv [L(B=B0dB) = B0dB]w/ boundary", sum_bnd(k))
        end if
        write (15,'(A)') " # Boundary conversion (Method 2)"
        if (trim(opname(k)) == "0") then
            call two("bndconv2 [L(B=0) = B0dB]   w/ boundary", sum_bnd2(k))
        else
            call two("bndconv2 [L(B=B0dB) = B0dB]w/ boundary", sum_bnd2(k))
        end if
    end if
    write (15,'(A)') " # Decomposition sum"
    tmp1 = sum_Qeta(k,1) + sum_Qeta(k,2) + sum_Qeta(k,3)                                                       ! :805, :833
    if (use_rchi_bc) tmp1 = tmp1 + sum_bnd(k)
    call two(lab("etaQ [L(", k, " = J F] w/  boundary"), tmp1)
    write (15,'(A)') " # wtheta integral"
    call two(lab("wtheta [L(", k, " = J F] w/  boundary"), sum_wtheta(k))
end do
close (15)
print *, "Time elapsed (sec): ", real(clk1 - clk0) / real(clk_rate)

contains

! "eta [L(B=0)    = dB] ..." / "eta [L(B=B0dB) = dB] ...": the label pieces of :788-838
function lab(head, k, tail) result(s)
    character(*), intent(in) :: head, tail
    integer, intent(in) :: k
    character(64) :: s
    if (trim(opname(k)) == "0") then
        s = head // "B=0)   " // tail
    else
        s = head // "B=B0dB)" // tail
    end if
end function

subroutine two(label, v)
    character(*), intent(in) :: label
    real(4), intent(in) :: v
    write (15,'(1X,A,A,ES16.9,A,ES16.9)') trim(label), " : ", v, " , ", v / sum_Q
end subroutine

! one legacy solve_elliptic on the device: strategy 1 / 2 as the r1- / r2-only cases of the current state machine
subroutine solve(strategy, r, max_iter, alpha, start, fresh, result)
    integer, intent(in) :: strategy, max_iter
    real(4), intent(in) :: r, alpha, start(nr,nz)
    logical, intent(in) :: fresh                  ! .false.: carry the field of the previous solve on (:545)
    real(4), intent(out) :: result(nr,nz)
    integer(c_int) :: st, flags
    flags = XEE_SOLVE_KEEP_PSI
    if (fresh) then
        call xee_check(xee_ensemble_upload(ens, XEE_F_BC_INIT, 0, 1, start), "upload start")
        flags = XEE_SOLVE_DEFAULT
    end if
    if (strategy == 1) then
        st = xee_ensemble_solve(ens, max_iter, 100, 1, 5, r, -1.0, alpha, flags)
    else
        st = xee_ensemble_solve(ens, max_iter, 100, 10, 5, -1.0, r, alpha, flags)
    end if
    call xee_check(st, "solve")
    call xee_check(xee_ensemble_records(ens, rec), "records")
    if (rec(1)%err_bits == 0) print *, "Elliptic Tools: Iteration success."
    if (iand(rec(1)%err_bits, 1) /= 0) print *, "Elliptic Tools: [Error] Max iteration reached."
    print *, "Relaxation uses ", rec(1)%iters, " steps. Final residue is ", rec(1)%err_now, "."
    call xee_check(xee_ensemble_download(ens, XEE_F_PSI, 0, 1, result), "download psi")
end subroutine

subroutine set_operator(k)
    integer, intent(in) :: k
    if (trim(opname(k)) == "0") then
        call xee_check(xee_ensemble_assemble_solver(ens, 0), "assemble L(B=0)")                                ! :553
    else
        call xee_check(xee_ensemble_upload(ens, XEE_F_SOLVER_B, 0, 1, sb_B0dB), "upload solver_b")
        call xee_check(xee_ensemble_assemble_solver(ens, 1), "assemble L(B0+dB)")                              ! :572
    end if
end subroutine

subroutine chi_solve(k, irhs, forcing, start, fresh)
    integer, intent(in) :: k, irhs
    real(4), intent(in) :: forcing(nr,nz), start(nr,nz)
    logical, intent(in) :: fresh
    character(16) :: tag
    call set_operator(k)
    call xee_check(xee_ensemble_upload(ens, XEE_F_FORCING, 0, 1, forcing), "upload f")
    call solve(strategy_rchi, r_rchi, max_iter_rchi, alpha_rchi, start, fresh, rchi)
    rchi_keep(:,:,k,irhs) = rchi
    call xee_check(xee_ensemble_cal_eta(ens), "cal_eta")                                                       ! :1129-1141
    call xee_check(xee_ensemble_ee_scalars(ens), "ee_scalars")                                                 ! :1073-1092
    call xee_check(xee_ensemble_records(ens, rec), "records")
    sum_Qeta(k, irhs) = rec(1)%sum_Qeta
    call xee_check(xee_ensemble_download(ens, XEE_F_ETA, 0, 1, eta), "download eta")
    tag = "[" // trim(opname(k)) // "_" // trim(rhsname(irhs)) // "]"
    call write_2Dfield(11, trim(output_folder)//"/eta-"//trim(tag)//"-A.bin", eta, nr-1, nz)
    call write_2Dfield(11, trim(output_folder)//"/rchi-"//trim(tag)//"-O.bin", rchi, nr, nz)
end subroutine

subroutine exchange(psi_a, chi, total)
    real(4), intent(in) :: psi_a(nr,nz), chi(nr,nz)
    real(4), intent(out) :: total
    if (legacy_records) then
        call xee_check(xee_exchange_conversion_legacy(psi_a, chi, rhoC_in, ra, za, rho, bnd, total, nr, nz), "exchange_conversion")
    else
        call xee_check(xee_exchange_conversion(psi_a, chi, rhoC_in, ra, za, rho, bnd, total, nr, nz), "exchange_conversion")
    end if
end subroutine

! like read_input but returns .false. at end of file instead of failing
logical function try_read_input(fd, line)
    integer, intent(in) :: fd
    character(256), intent(out) :: line
    character(256) :: raw
    integer :: cut, ios, n
    try_read_input = .false.
    line = ''
    do
        read (fd, '(a)', iostat=ios) raw
        if (ios /= 0) return
        cut = index(raw, '//')
        if (cut > 0) raw(cut:) = ' '
        n = len_trim(raw)
        if (n > 0) then
            if (raw(n:n) == achar(13)) raw(n:n) = ' '
        end if
        if (len_trim(raw) > 0) exit
    end do
    line = raw
    try_read_input = .true.
end function

end program xee_old_diagnose
