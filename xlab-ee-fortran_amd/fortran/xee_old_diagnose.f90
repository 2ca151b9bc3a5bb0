! xee_old_diagnose.f90 -- Fortran 2008 host of the LEGACY program on the MI355X: `xee-old-diagnose < diag.txt`.
!
! Keeps the surface of src/old-diagnose/diagnose.f90: the positional stdin config with // comments (:76-167, SURVEY.md
! Appendix A.2), the raw float32 input files, every output file name and shape (:389-770) and efficiency.txt with all its
! labels (:779-841, what xtt-lib-python/XEffReader.py parses).  The stage order itself -- source terms, Stage I -> II -> III,
! integral check, exchange conversion, efficiency.txt -- is ONE libxee call, xee_legacy_run (csrc/xee_legacy.hip; the
! Python batch driver legacy_diagnose.py calls the same entry): this program reads the config and the input files, hands
! them over, prints the solver lines and lets xee_legacy_write_outputs write the files.  No CPU fallback.
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
character(256) :: mode_str, word(4), buffer, input_folder, output_folder, A_file, B_file, C_file, Q_file, F_file, &
&                 rpsi_bc_file, rchi_bc_file, yes_or_no
integer :: nr, nz, geometry, density_mode, op_mode, i, k, ndev, n_solves
integer :: strategy_rpsi, strategy_rchi, max_iter_rpsi, max_iter_rchi
real(4) :: testing_dt, domain(4), Lr(2), Lz(2), planet_radius, r_rpsi, r_rchi, alpha_rpsi, alpha_rchi
logical :: use_rpsi_bc, use_rchi_bc, legacy_records
real(4), allocatable, target :: rhoA_in(:,:), rhoB_in(:,:), rhoC_in(:,:), Q_in(:,:), F_in(:,:), rpsi_bc(:,:), rchi_bc(:,:)
real(4), allocatable :: record(:), err_now(:)
integer(c_int), allocatable :: iters(:)
type(c_ptr) :: res, p_psi_bc, p_chi_bc
type(xee_legacy_config) :: cfg
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

allocate(rhoA_in(nr,nz), rhoB_in(nr,nz), rhoC_in(nr,nz), Q_in(nr-1,nz-1), F_in(nr-1,nz-1))
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
p_psi_bc = c_null_ptr
p_chi_bc = c_null_ptr
if (use_rpsi_bc) then
    allocate(rpsi_bc(nr,nz)); call read_2Dfield(15, trim(input_folder)//"/"//rpsi_bc_file, rpsi_bc, nr, nz)
    p_psi_bc = c_loc(rpsi_bc)
end if
if (use_rchi_bc) then
    allocate(rchi_bc(nr,nz)); call read_2Dfield(15, trim(input_folder)//"/"//rchi_bc_file, rchi_bc, nr, nz)
    p_chi_bc = c_loc(rchi_bc)
end if
print *, "Initialization complete."

! ---- the legacy program's stages (old-diagnose/diagnose.f90:283-772), one call ----------------------------------------
cfg%geometry = geometry;  cfg%density_mode = density_mode;  cfg%domain = domain;  cfg%testing_dt = testing_dt
cfg%op_mode = op_mode
cfg%psi_strategy = strategy_rpsi;  cfg%psi_r = r_rpsi;  cfg%psi_max_iter = max_iter_rpsi;  cfg%psi_alpha = alpha_rpsi
cfg%chi_strategy = strategy_rchi;  cfg%chi_r = r_rchi;  cfg%chi_max_iter = max_iter_rchi;  cfg%chi_alpha = alpha_rchi
cfg%legacy_int = merge(1, 0, legacy_records)      ! the program's INTEGER r, dr, dz in the exchange conversion (:1146)
cfg%chained = 1                                   ! first guesses carried from solve to solve, as the program does (:545)
cfg%device = -1
cfg%first_case_id = 0
print *, "Solving rpsi, CHI (", merge("with   ", "without", use_rchi_bc), " boundary condition), integral check..."
call xee_check(xee_legacy_run(res, cfg, nr, nz, 1, rhoA_in, rhoB_in, rhoC_in, Q_in, F_in, p_psi_bc, p_chi_bc), "legacy_run")
call xee_check(xee_legacy_relax(res, 0, n_solves, c_null_ptr, c_null_ptr, 0), "relax")
allocate(iters(n_solves), err_now(n_solves))
call xee_check(xee_legacy_relax_get(res, 0, n_solves, iters, err_now, n_solves), "relax")
do k = 1, n_solves                                ! the lines solve_elliptic's callers print (:452-461 ...)
    print *, "Relaxation uses ", iters(k), " steps. Final residue is ", err_now(k), "."
end do
call system_clock(clk1)
call xee_check(xee_legacy_write_outputs(res, 0, trim(output_folder)//c_null_char, real(clk1 - clk0, c_double) / real(clk_rate, c_double)), &
&              "write_outputs")
call xee_check(xee_legacy_destroy(res), "legacy_destroy")
print *, "Time elapsed (sec): ", real(clk1 - clk0) / real(clk_rate)

contains

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
