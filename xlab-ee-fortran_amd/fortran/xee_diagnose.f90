! xee_diagnose.f90 -- Fortran 2008 host of the MI355X path: `xee-diagnose < diag.txt`.
!
! Keeps the reference program's surface (src/diagnose/main.f90 and its includes): the positional
! stdin config with // comments (read-input.f90:1-118, SURVEY.md Appendix A.1), the raw float32
! input files (initialize-variables.f90:33-42), the debug_mode_1/2 sentinel files (main.f90:13-21),
! the output file names and shapes (initialize-variables.f90:127-129, diagnose.f90:18-27,45-54) and
! result.txt (write-output.f90:1-3).  The numerical stages run on the GPU through libxee's
! device-resident ensemble API (one member), so A/B/C/bc are uploaded once and only the requested
! output fields come back:
!     forcing (A7) -> solver coefficients (K1) -> cal_coe (K2) -> solve_elliptic (K3+K4)
!     -> cal_eta | cal_uw (K5/K6) -> [EE scalars (K8)]
!
! Backward-compatible extension (SURVEY.md 8a row A10): ONE optional extra line after the solver
! line names a heating file Q on B cells (nr-1,nz-1).  When present (DYNAMIC_EFFICIENCY only) the
! program also writes efficiency.txt with sum Q and, per operator variant, sum(Q*eta) and the
! efficiency sum(Q*eta)/sum(Q), under the legacy labels XEffReader.py keys on
! (old-diagnose/diagnose.f90:779-841).  The reference stops reading before that line, so old
! configs and the reference itself are unaffected.
program xee_diagnose
use, intrinsic :: iso_c_binding
use xee_c_api
use field_tools
use read_input_tools
use message_tools
implicit none

integer, parameter :: stdin = 5, ERROR_INPUT = 1
integer, parameter :: CYLINDRICAL_MODE = 0, SPHERICAL_MODE = 1
integer, parameter :: DIAGPARAM_DYNAMIC_EFFICIENCY = 0, DIAGPARAM_SECONDARY_CIRCULATION = 1, DIAGPARAM_NONE = 2
integer, parameter :: DENSITY_NORMAL = 0, DENSITY_BOUSSINESQ = 1
integer, parameter :: BARO_ALL = 2, BAROCLINIC = 1, BAROTROPIC = 0

character(256) :: mode_str, word(4), buffer, input_folder, output_folder, A_file, B_file, C_file, &
&                 forcing_file, bc_init_file, Q_file
integer :: nr, nz, diag_param, geometry, density_mode, operator_complexity, debug_mode
integer :: i, saved_max_iter, ndev, variant, nvariants, variants(2)
real(4) :: domain(4), Lr(2), Lz(2), planet_radius, saved_r1, saved_r2, alpha_strf, dr, dz
real(4), allocatable :: fld(:,:), ra(:), za(:), rcuva(:), exner(:), rho(:), sin_table(:)
real(4), allocatable :: out_O(:,:), out_A(:,:), out_C(:,:), out_sA(:,:), out_B(:,:), out_sC(:,:), Q_in(:,:)
logical :: file_exists, have_Q
type(c_ptr) :: ens
type(xee_case_record) :: rec(1)
integer(c_int) :: st
integer(8) :: clk0, clk1, clk_rate
real(4) :: sum_Q, sum_Qeta(2)
character(16) :: vname

print *, "Dynamic Efficiency Diagnose Program (MI355X / libxee host)"

debug_mode = 0                                       ! main.f90:13-21
inquire(file="./debug_mode_1", exist=file_exists); if (file_exists) debug_mode = 1
inquire(file="./debug_mode_2", exist=file_exists); if (file_exists) debug_mode = 2

! ---- config (read-input.f90) -------------------------------------------------------------------
call read_input(stdin, mode_str)
do i = 1, size(word)
    if (split_line(mode_str, word(i), "-") /= 0 .and. i /= size(word)) then
        call error_msg("INIT", ERROR_INPUT, "There should be 4 inputs separated by dashes.")
        stop
    end if
end do

select case (trim(word(1)))
case ('DYNAMIC_EFFICIENCY');    diag_param = DIAGPARAM_DYNAMIC_EFFICIENCY
case ('SECONDARY_CIRCULATION'); diag_param = DIAGPARAM_SECONDARY_CIRCULATION
case ('NONE');                  diag_param = DIAGPARAM_NONE
case default
    call error_msg("INIT", ERROR_INPUT, "Unknown Mode [" // trim(word(1)) // "]"); stop
end select
select case (trim(word(2)))
case ('CYLINDRICAL'); geometry = CYLINDRICAL_MODE
case ('SPHERICAL');   geometry = SPHERICAL_MODE
case default
    call error_msg("INIT", ERROR_INPUT, "Unknown Mode [" // trim(word(2)) // "]"); stop
end select
select case (trim(word(3)))
case ('DENSITY_NORMAL');     density_mode = DENSITY_NORMAL
case ('DENSITY_BOUSSINESQ'); density_mode = DENSITY_BOUSSINESQ
case default
    call error_msg("INIT", ERROR_INPUT, "Unknown Mode [" // trim(word(3)) // "]"); stop
end select
select case (trim(word(4)))
case ('BARO_ALL');   operator_complexity = BARO_ALL
case ('BAROCLINIC'); operator_complexity = BAROCLINIC
case ('BAROTROPIC'); operator_complexity = BAROTROPIC
case default
    call error_msg("INIT", ERROR_INPUT, "Unknown Mode [" // trim(word(4)) // "]"); stop
end select

domain = 0.0
call read_input(stdin, buffer)
if (geometry == CYLINDRICAL_MODE) then
    read (buffer, *) Lr(1), Lr(2), Lz(1), Lz(2)
    if (Lr(2) <= Lr(1)) call error_msg("INIT", ERROR_INPUT, "Domain size in radial direction must be positive.")
    domain = [Lr(1), Lr(2), Lz(1), Lz(2)]
else
    read (buffer, *) planet_radius, Lz(1), Lz(2)
    domain(1:3) = [planet_radius, Lz(1), Lz(2)]
end if
if (Lz(2) <= Lz(1)) call error_msg("INIT", ERROR_INPUT, "Domain size in z direction must be positive.")

call read_input(stdin, buffer); read (buffer, *) nr, nz
call read_input(stdin, input_folder)
call read_input(stdin, output_folder)
call read_input(stdin, A_file)
call read_input(stdin, B_file)
call read_input(stdin, C_file)
if (diag_param == DIAGPARAM_SECONDARY_CIRCULATION) call read_input(stdin, forcing_file)
call read_input(stdin, bc_init_file)
call read_input(stdin, buffer)
read (buffer, *) saved_r1, saved_r2, saved_max_iter, alpha_strf
have_Q = try_read_input(stdin, Q_file) .and. diag_param == DIAGPARAM_DYNAMIC_EFFICIENCY

print *, "----- Diagnose Input -----"
print *, "Diagnose parameter: ", diag_param
print *, "Geometry: ", geometry
print *, "Density distribution: ", density_mode
print *, "Operator complexity: ", operator_complexity
if (geometry == CYLINDRICAL_MODE) then
    print *, "Lr:", Lr(1), Lr(2)
else
    print *, "Using spherical mode, domain is forced to be global."
    print *, "Planet Radius: ", planet_radius
end if
print *, "Lz:", Lz(1), Lz(2)
print *, "nr:", nr, ", nz:", nz
print *, "Input folder:  ", trim(input_folder)
print *, "Output folder: ", trim(output_folder)
print *, "A file:        ", trim(A_file)
print *, "B file:        ", trim(B_file)
print *, "C file:        ", trim(C_file)
if (diag_param == DIAGPARAM_SECONDARY_CIRCULATION) print *, "forcing file:  ", trim(forcing_file)
print *, "bc_init file:  ", trim(bc_init_file)
if (have_Q) print *, "Q file:        ", trim(Q_file)
print *, "absolute, relative residue, iter, alpha: ", saved_r1, saved_r2, saved_max_iter, alpha_strf
print *, "--------------------------"
print *, "Read input complete."

! ---- device + inputs (initialize-variables.f90) ---------------------------------------------------
call xee_check(xee_device_count(ndev), "device_count")
if (ndev < 1) then
    call error_msg("INIT", ERROR_INPUT, "no MI355X visible: this program has no CPU fallback")
    error stop 1
end if
allocate(fld(nr,nz), ra(nr), za(nz), rcuva(nr), exner(nz), rho(nz), sin_table(nr))
allocate(out_O(nr,nz), out_A(nr-1,nz), out_C(nr,nz-1), out_sA(nr-1,nz-2), out_B(nr-1,nz-1), out_sC(nr-2,nz-1))

call xee_check(xee_ensemble_create(ens, nr, nz, 1, -1), "ensemble_create")
call xee_check(xee_profiles(geometry, density_mode, domain, nr, nz, ra, za, rcuva, exner, rho, sin_table, dr, dz), "profiles")
call xee_check(xee_ensemble_set_profiles(ens, ra, za, rcuva, rho, exner, dr, dz), "set_profiles")
print *, "Geometry complete."

call read_2Dfield(15, trim(input_folder)//"/"//A_file, fld, nr, nz)
call xee_check(xee_ensemble_upload(ens, XEE_F_A, 0, 1, fld), "upload A")
call read_2Dfield(15, trim(input_folder)//"/"//B_file, fld, nr, nz)
call xee_check(xee_ensemble_upload(ens, XEE_F_B, 0, 1, fld), "upload B")
call read_2Dfield(15, trim(input_folder)//"/"//C_file, fld, nr, nz)
call xee_check(xee_ensemble_upload(ens, XEE_F_C, 0, 1, fld), "upload C")
call read_2Dfield(15, trim(input_folder)//"/"//bc_init_file, fld, nr, nz)
call xee_check(xee_ensemble_upload(ens, XEE_F_BC_INIT, 0, 1, fld), "upload bc_init")
if (diag_param == DIAGPARAM_SECONDARY_CIRCULATION) then
    call read_2Dfield(15, trim(input_folder)//"/"//forcing_file, fld, nr, nz)
    call xee_check(xee_ensemble_upload(ens, XEE_F_FORCING, 0, 1, fld), "upload forcing")
else if (diag_param == DIAGPARAM_DYNAMIC_EFFICIENCY) then
    call xee_check(xee_ensemble_forcing_from_B(ens), "forcing_from_B")       ! forcing_in = - rhoB_in
else
    fld = 0.0                                                                 ! quirk 13: NONE -> zero RHS
    call xee_check(xee_ensemble_upload(ens, XEE_F_FORCING, 0, 1, fld), "upload zero forcing")
end if
if (have_Q) then
    allocate(Q_in(nr-1,nz-1))
    call read_2Dfield(15, trim(input_folder)//"/"//Q_file, Q_in, nr-1, nz-1)
    call xee_check(xee_ensemble_upload(ens, XEE_F_Q, 0, 1, Q_in), "upload Q")
end if
print *, "Initialization complete."

nvariants = 0                                        ! diagnose.f90:4,31
if (operator_complexity == BAROTROPIC .or. operator_complexity == BARO_ALL) then
    nvariants = nvariants + 1; variants(nvariants) = XEE_BAROTROPIC
end if
if (operator_complexity == BAROCLINIC .or. operator_complexity == BARO_ALL) then
    nvariants = nvariants + 1; variants(nvariants) = XEE_BAROCLINIC
end if

call system_clock(clk0, clk_rate)
sum_Q = 0.0; sum_Qeta = 0.0
do variant = 1, nvariants
    if (variants(variant) == XEE_BAROTROPIC) then
        vname = "BAROTROPIC"
        print *, "Solving CHI with L(A,B=0,C) = -B"
    else
        vname = "BAROCLINIC"
        print *, "Solving CHI with L(A,B,C) = -B"
    end if
    call xee_check(xee_ensemble_assemble(ens, variants(variant)), "assemble")
    if (variant == 1) then                           ! initialize-variables.f90:127-129
        call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_A, 0, 1, out_sA), "download solver A")
        call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_B, 0, 1, out_B), "download solver B")
        call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_C, 0, 1, out_sC), "download solver C")
        call write_2Dfield(11, trim(output_folder)//"/solver_a-sA.bin", out_sA, nr-1, nz-2)
        call write_2Dfield(11, trim(output_folder)//"/solver_b-B.bin", out_B, nr-1, nz-1)
        call write_2Dfield(11, trim(output_folder)//"/solver_c-sC.bin", out_sC, nr-2, nz-1)
        print *, "Solver complete."
    end if

    st = xee_ensemble_solve(ens, saved_max_iter, 100, 10, 5, saved_r1, saved_r2, alpha_strf, &
    &                       merge(XEE_SOLVE_TRACE, XEE_SOLVE_DEFAULT, debug_mode == 2))
    if (st == XEE_ERR_INVALID_CRITERIA) then
        print *, "ERROR: [check_abs_err] and [check_rel_err] cannot both be non-positive."
        stop
    end if
    call xee_check(st, "solve")
    call xee_check(xee_ensemble_records(ens, rec), "records")
    call print_judge(rec(1)%err_bits)
    print *, "Relaxation uses ", rec(1)%iters, " steps. Final residue is ", rec(1)%err_now, ",", rec(1)%ratio

    call xee_check(xee_ensemble_download(ens, XEE_F_PSI, 0, 1, out_O), "download psi")
    if (diag_param == DIAGPARAM_DYNAMIC_EFFICIENCY) then
        call xee_check(xee_ensemble_cal_eta(ens), "cal_eta")
        call xee_check(xee_ensemble_download(ens, XEE_F_ETA, 0, 1, out_A), "download eta")
        call write_2Dfield(11, trim(output_folder)//"/eta-["//trim(vname)//"]-A.bin", out_A, nr-1, nz)
        call write_2Dfield(11, trim(output_folder)//"/rchi-["//trim(vname)//"]-O.bin", out_O, nr, nz)
        if (have_Q) then
            call xee_check(xee_ensemble_ee_scalars(ens), "ee_scalars")
            call xee_check(xee_ensemble_records(ens, rec), "records")
            sum_Q = rec(1)%sum_Q
            sum_Qeta(variant) = rec(1)%sum_Qeta
        end if
    else if (diag_param == DIAGPARAM_SECONDARY_CIRCULATION) then
        call xee_check(xee_ensemble_cal_uw(ens), "cal_uw")
        call xee_check(xee_ensemble_download(ens, XEE_F_W, 0, 1, out_A), "download w")
        call xee_check(xee_ensemble_download(ens, XEE_F_U, 0, 1, out_C), "download u")
        call write_2Dfield(11, trim(output_folder)//"/w-["//trim(vname)//"]-A.bin", out_A, nr-1, nz)
        call write_2Dfield(11, trim(output_folder)//"/u-["//trim(vname)//"]-C.bin", out_C, nr, nz-1)
        call write_2Dfield(11, trim(output_folder)//"/rpsi-["//trim(vname)//"]-O.bin", out_O, nr, nz)
    end if
end do
call system_clock(clk1)
print *, "Diagnose complete."

open (unit=15, file=trim(output_folder)//"/result.txt", action="write", status="replace")   ! write-output.f90
write (15,*) "Time elapsed (sec) : ", real(clk1 - clk0) / real(clk_rate)
close (15)

if (have_Q) then                                     ! efficiency.txt, old-diagnose/diagnose.f90:779-841
    open (unit=15, file=trim(output_folder)//"/efficiency.txt", action="write", status="replace")
    write (15,'(A,ES16.9)') " Time elapsed (sec)                          : ", real(clk1 - clk0) / real(clk_rate)
    write (15,'(A,ES16.9)') " sum Q                                       : ", sum_Q
    do variant = 1, nvariants
        write (15,'(A)') " # Internal efficiency"
        if (variants(variant) == XEE_BAROTROPIC) then
            write (15,'(A,ES16.9,A,ES16.9)') " eta [L(B=0)    = B0]     wo/ boundary : ", sum_Qeta(variant), " , ", sum_Qeta(variant) / sum_Q
        else
            write (15,'(A,ES16.9,A,ES16.9)') " eta [L(B=B0dB) = B0]     wo/ boundary : ", sum_Qeta(variant), " , ", sum_Qeta(variant) / sum_Q
        end if
    end do
    close (15)
end if

call xee_check(xee_ensemble_destroy(ens), "ensemble_destroy")

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

! the judge_error texts of xtt-lib-fortran/elliptic_tools.f90:333-358
subroutine print_judge(err)
    integer, intent(in) :: err
    if (err == 0) print *, "Elliptic Tools: Iteration success."
    if (iand(err, 1) /= 0) print *, "Elliptic Tools: [Error] Max iteration reached."
    if (iand(err, 2) /= 0) print *, "Elliptic Tools: [Error] Iteration explodes."
    if (err /= 0 .and. iand(err, 3) == 0) print *, "Elliptic Tools: Unknown error code ", err
end subroutine

end program xee_diagnose
