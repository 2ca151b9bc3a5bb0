! xee_ensemble.f90 -- Fortran 2008 ENSEMBLE host of the MI355X path: `xee-ensemble ensemble.txt`, one process per GPU.
!
! The reference diagnoses one case per run (src/diagnose/main.f90); the per-run solves of src/diagnose/diagnose.f90:4-55
! are independent between runs, so an ensemble of cases that share one grid and one mode line is partitioned in
! consecutive blocks over the ranks (static block partition, SURVEY.md 8e), every rank keeps its block resident in
! ONE device ensemble (members = cases: all of them advance in the same sweep launches and stop on their own criterion),
! writes its cases' output files exactly as xee-diagnose would, and the 128-byte case records are all-gathered over
! RCCL through libxee's C-ABI (xee_gather_scalars); rank 0 writes ensemble_summary.txt.
!
! Config: the file named by the first command-line argument, else by XEE_CONFIG, else stdin (a launcher such as Open MPI
! forwards stdin to rank 0 ONLY: with more than one rank give the file by name).  The reference's line reader: // comments,
! blank lines skipped:
!     TYPE-GEOMETRY-DENSITY-OPERATOR           as diag.txt line 1 (read-input.f90:2-51); TYPE in DYNAMIC_EFFICIENCY | NONE
!     Lr1 Lr2 Lz1 Lz2 | planet_radius Lz1 Lz2  (read-input.f90:55-76)
!     nr nz
!     r1 r2 max_iter alpha                     (read-input.f90:88-89)
!     n_cases
!     <case directory> x n_cases               each holds A.bin B.bin C.bin bc_init.bin [Q.bin]; outputs go there too
! Ranks: XEE_RANK / XEE_WORLD (else OMPI_COMM_WORLD_RANK/SIZE, else RANK/WORLD_SIZE; default 0/1), device = XEE_LOCAL_RANK
! (else LOCAL_RANK, else rank).  Rendezvous: rank 0 writes a time stamp and the 128-byte RCCL id to
! <first case directory>/../.xee_comm_id[.<job>] (override: XEE_COMM_FILE; <job> = XEE_JOB_ID, else the launcher's
! OMPI_MCA_ess_base_jobid / SLURM_JOB_ID / TORCHELASTIC_RUN_ID when set), the others wait for a file whose stamp is not
! older than their own start (a file left behind by a run that died is ignored, not joined), and xee_comm_create itself
! gives up after a time-out (XEE_COMM_TIMEOUT_S, default 120) instead of hanging when a rank never arrives.
! XEE_ENSEMBLE_GATHER=files replaces the RCCL gather by per-rank record files next to the rendezvous file: the rehearsal of
! the multi-rank path on ONE GPU (RCCL refuses two ranks on one device).
program xee_ensemble
use, intrinsic :: iso_c_binding
use xee_c_api
use field_tools
use read_input_tools
use message_tools
implicit none

integer :: stdin = 5
character(256) :: mode_str, word(4), buffer, comm_file, job_id, rec_file
integer(8) :: t_start, t_file
real(8) :: comm_timeout
logical :: gather_by_files
character(256), allocatable :: case_dir(:)
integer :: nr, nz, geometry, density_mode, operator_complexity, n_cases, rank, world, device, lo, hi, per, n_local
integer :: i, k, m, saved_max_iter, ndev, variant, nvariants, variants(2), ios
real(4) :: domain(4), Lr(2), Lz(2), planet_radius, saved_r1, saved_r2, alpha_strf, dr, dz
real(4), allocatable :: fld(:,:), ra(:), za(:), rcuva(:), exner(:), rho(:), sin_table(:)
real(4), allocatable :: out_O(:,:), out_A(:,:), out_sA(:,:), out_B(:,:), out_sC(:,:), Q_in(:,:)
logical :: file_exists, have_Q, dynamic_eff
type(c_ptr) :: ens, comm
type(xee_case_record), allocatable :: rec(:), mine(:), everyone(:)
character(kind=c_char) :: comm_id(128)
integer(c_int) :: st
integer(8) :: clk0, clk1, clk_rate
character(16) :: vname
character(64) :: val

call env_int("XEE_RANK", "OMPI_COMM_WORLD_RANK", "RANK", 0, rank)
call env_int("XEE_WORLD", "OMPI_COMM_WORLD_SIZE", "WORLD_SIZE", 1, world)
call env_int("XEE_LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "LOCAL_RANK", rank, device)

t_start = wall_seconds()
call get_command_argument(1, buffer, status=ios)                          ! the config: argument 1, XEE_CONFIG, or stdin
if (ios /= 0 .or. len_trim(buffer) == 0) call get_environment_variable("XEE_CONFIG", buffer, status=ios)
if (ios == 0 .and. len_trim(buffer) > 0) then
    open (newunit=stdin, file=trim(buffer), action="read", status="old", iostat=ios)
    if (ios /= 0) then
        call error_msg("INIT", 1, "cannot open the config file " // trim(buffer)); error stop 1
    end if
else if (world > 1 .and. rank > 0) then
    print *, "xee-ensemble: rank ", rank, " reads its config from stdin -- most launchers forward stdin to rank 0 only;"
    print *, "              name the file as the first argument (or XEE_CONFIG) instead."
end if

call read_input(stdin, mode_str)
do i = 1, size(word)
    if (split_line(mode_str, word(i), "-") /= 0 .and. i /= size(word)) then
        call error_msg("INIT", 1, "There should be 4 inputs separated by dashes."); stop
    end if
end do
select case (trim(word(1)))
case ('DYNAMIC_EFFICIENCY'); dynamic_eff = .true.
case ('NONE');               dynamic_eff = .false.
case default
    call error_msg("INIT", 1, "ensemble mode diagnoses DYNAMIC_EFFICIENCY or NONE, not [" // trim(word(1)) // "]"); stop
end select
geometry = merge(0, 1, trim(word(2)) == 'CYLINDRICAL')
if (geometry == 1 .and. trim(word(2)) /= 'SPHERICAL') then
    call error_msg("INIT", 1, "Unknown Mode [" // trim(word(2)) // "]"); stop
end if
density_mode = merge(0, 1, trim(word(3)) == 'DENSITY_NORMAL')
if (density_mode == 1 .and. trim(word(3)) /= 'DENSITY_BOUSSINESQ') then
    call error_msg("INIT", 1, "Unknown Mode [" // trim(word(3)) // "]"); stop
end if
select case (trim(word(4)))
case ('BARO_ALL');   operator_complexity = 2
case ('BAROCLINIC'); operator_complexity = 1
case ('BAROTROPIC'); operator_complexity = 0
case default
    call error_msg("INIT", 1, "Unknown Mode [" // trim(word(4)) // "]"); stop
end select
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
call read_input(stdin, buffer); read (buffer, *) saved_r1, saved_r2, saved_max_iter, alpha_strf
call read_input(stdin, buffer); read (buffer, *) n_cases
if (n_cases < 1) then
    call error_msg("INIT", 1, "n_cases must be positive"); stop
end if
allocate(case_dir(n_cases))
do k = 1, n_cases
    call read_input(stdin, case_dir(k))
end do

! static block partition (SURVEY.md 8e): rank g owns cases [g*per, min(N, (g+1)*per)), 0-based
per = (n_cases + world - 1) / world
lo = min(n_cases, rank * per)
hi = min(n_cases, lo + per)
n_local = hi - lo
if (rank == 0) print *, "xee-ensemble: ", n_cases, " cases on ", world, " rank(s), ", per, " per rank"

call xee_check(xee_device_count(ndev), "device_count")
if (ndev < 1) then
    call error_msg("INIT", 1, "no MI355X visible: this program has no CPU fallback"); error stop 1
end if
device = mod(device, ndev)

comm = c_null_ptr
gather_by_files = .false.
if (world > 1 .or. env_set("XEE_ENSEMBLE_FORCE_GATHER")) then          ! the collective's rendezvous
    call get_environment_variable("XEE_COMM_FILE", comm_file, status=ios)
    if (ios /= 0 .or. len_trim(comm_file) == 0) then
        comm_file = trim(case_dir(1)) // "/../.xee_comm_id"
        call first_env(["XEE_JOB_ID              ", "OMPI_MCA_ess_base_jobid ", "SLURM_JOB_ID            ", "TORCHELASTIC_RUN_ID     "], job_id)
        if (len_trim(job_id) > 0) comm_file = trim(comm_file) // "." // trim(job_id)
    end if
    call get_environment_variable("XEE_ENSEMBLE_GATHER", buffer, status=ios)
    gather_by_files = ios == 0 .and. trim(buffer) == "files"
    comm_timeout = 120.0d0
    call get_environment_variable("XEE_COMM_TIMEOUT_S", buffer, status=ios)
    if (ios == 0 .and. len_trim(buffer) > 0) read (buffer, *, iostat=ios) comm_timeout
    call xee_check(xee_comm_set_timeout(comm_timeout), "comm_set_timeout")
    if (rank == 0) then
        comm_id = c_null_char
        if (.not. gather_by_files) call xee_check(xee_comm_unique_id(comm_id), "comm_unique_id")
        open (unit=17, file=trim(comm_file)//".tmp", access="stream", form="unformatted", status="replace")
        write (17) wall_seconds(), comm_id
        close (17)
        call execute_command_line("mv -f '" // trim(comm_file) // ".tmp' '" // trim(comm_file) // "'")
    else
        file_exists = .false.
        do i = 1, int(comm_timeout * 10.0d0) + 1
            inquire(file=trim(comm_file), exist=file_exists)
            if (file_exists) then
                ! a file from THIS launch carries a stamp no older than this rank's own start (a few seconds of slack for
                ! start-up skew); anything older was left behind by a run that died: keep waiting for rank 0 to replace it
                open (unit=17, file=trim(comm_file), access="stream", form="unformatted", status="old", iostat=ios)
                if (ios == 0) then
                    read (17, iostat=ios) t_file, comm_id
                    close (17)
                end if
                file_exists = ios == 0 .and. t_file >= t_start - 5
                if (file_exists) exit
            end if
            call execute_command_line("sleep 0.1")
        end do
        if (.not. file_exists) then
            call error_msg("INIT", 1, "rank 0 never wrote a fresh " // trim(comm_file)); error stop 1
        end if
    end if
    if (.not. gather_by_files) then
        call xee_check(xee_comm_create(comm, comm_id, rank, world, device), "comm_create")
        if (rank == 0) call execute_command_line("rm -f '" // trim(comm_file) // "'")     ! (everyone has joined: create is collective)
    end if
end if

allocate(mine(per), everyone(per * world))
mine(:)%case_id = -1
call system_clock(clk0, clk_rate)
if (n_local > 0) then
    allocate(fld(nr,nz), ra(nr), za(nz), rcuva(nr), exner(nz), rho(nz), sin_table(nr), rec(n_local))
    allocate(out_O(nr,nz), out_A(nr-1,nz), out_sA(nr-1,nz-2), out_B(nr-1,nz-1), out_sC(nr-2,nz-1), Q_in(nr-1,nz-1))
    call xee_check(xee_ensemble_create(ens, nr, nz, n_local, device), "ensemble_create")
    call xee_check(xee_ensemble_set_case_ids(ens, lo), "set_case_ids")
    call xee_check(xee_profiles(geometry, density_mode, domain, nr, nz, ra, za, rcuva, exner, rho, sin_table, dr, dz), "profiles")
    call xee_check(xee_ensemble_set_profiles(ens, ra, za, rcuva, rho, exner, dr, dz), "set_profiles")
    inquire(file=trim(case_dir(lo+1))//"/Q.bin", exist=have_Q)
    have_Q = have_Q .and. dynamic_eff
    do m = 0, n_local - 1                             ! inputs resident on the device before any solve
        call read_2Dfield(15, trim(case_dir(lo+m+1))//"/A.bin", fld, nr, nz)
        call xee_check(xee_ensemble_upload(ens, XEE_F_A, m, 1, fld), "upload A")
        call read_2Dfield(15, trim(case_dir(lo+m+1))//"/B.bin", fld, nr, nz)
        call xee_check(xee_ensemble_upload(ens, XEE_F_B, m, 1, fld), "upload B")
        call read_2Dfield(15, trim(case_dir(lo+m+1))//"/C.bin", fld, nr, nz)
        call xee_check(xee_ensemble_upload(ens, XEE_F_C, m, 1, fld), "upload C")
        call read_2Dfield(15, trim(case_dir(lo+m+1))//"/bc_init.bin", fld, nr, nz)
        call xee_check(xee_ensemble_upload(ens, XEE_F_BC_INIT, m, 1, fld), "upload bc_init")
        if (.not. dynamic_eff) then
            fld = 0.0                                 ! quirk 13: NONE -> zero RHS
            call xee_check(xee_ensemble_upload(ens, XEE_F_FORCING, m, 1, fld), "upload zero forcing")
        end if
        if (have_Q) then
            call read_2Dfield(15, trim(case_dir(lo+m+1))//"/Q.bin", Q_in, nr-1, nz-1)
            call xee_check(xee_ensemble_upload(ens, XEE_F_Q, m, 1, Q_in), "upload Q")
        end if
    end do
    if (dynamic_eff) call xee_check(xee_ensemble_forcing_from_B(ens), "forcing_from_B")

    nvariants = 0                                     ! diagnose.f90:4,31
    if (operator_complexity == 0 .or. operator_complexity == 2) then
        nvariants = nvariants + 1; variants(nvariants) = XEE_BAROTROPIC
    end if
    if (operator_complexity == 1 .or. operator_complexity == 2) then
        nvariants = nvariants + 1; variants(nvariants) = XEE_BAROCLINIC
    end if
    do variant = 1, nvariants
        vname = merge("BAROTROPIC", "BAROCLINIC", variants(variant) == XEE_BAROTROPIC)
        call xee_check(xee_ensemble_assemble(ens, variants(variant)), "assemble")
        st = xee_ensemble_solve(ens, saved_max_iter, 100, 10, 5, saved_r1, saved_r2, alpha_strf, XEE_SOLVE_DEFAULT)
        if (st == XEE_ERR_INVALID_CRITERIA) then
            print *, "ERROR: [check_abs_err] and [check_rel_err] cannot both be non-positive."
            stop
        end if
        call xee_check(st, "solve")
        if (dynamic_eff) then
            call xee_check(xee_ensemble_cal_eta(ens), "cal_eta")
            if (have_Q) call xee_check(xee_ensemble_ee_scalars(ens), "ee_scalars")
        end if
        call xee_check(xee_ensemble_records(ens, rec), "records")
        do m = 0, n_local - 1
            print '(A,I0,A,A,A,I0,A,ES14.7,A,ES14.7)', " case ", lo + m, " [", trim(vname), "]: Relaxation uses ", rec(m+1)%iters, &
            &     " steps. Final residue is ", rec(m+1)%err_now, " , ", rec(m+1)%ratio
            if (variant == 1) then                    ! initialize-variables.f90:127-129
                call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_A, m, 1, out_sA), "download solver A")
                call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_B, m, 1, out_B), "download solver B")
                call xee_check(xee_ensemble_download(ens, XEE_F_SOLVER_C, m, 1, out_sC), "download solver C")
                call write_2Dfield(11, trim(case_dir(lo+m+1))//"/solver_a-sA.bin", out_sA, nr-1, nz-2)
                call write_2Dfield(11, trim(case_dir(lo+m+1))//"/solver_b-B.bin", out_B, nr-1, nz-1)
                call write_2Dfield(11, trim(case_dir(lo+m+1))//"/solver_c-sC.bin", out_sC, nr-2, nz-1)
            end if
            if (dynamic_eff) then                     ! diagnose.f90:18-21,45-48
                call xee_check(xee_ensemble_download(ens, XEE_F_PSI, m, 1, out_O), "download psi")
                call xee_check(xee_ensemble_download(ens, XEE_F_ETA, m, 1, out_A), "download eta")
                call write_2Dfield(11, trim(case_dir(lo+m+1))//"/eta-["//trim(vname)//"]-A.bin", out_A, nr-1, nz)
                call write_2Dfield(11, trim(case_dir(lo+m+1))//"/rchi-["//trim(vname)//"]-O.bin", out_O, nr, nz)
            end if
        end do
    end do
    mine(1:n_local) = rec(1:n_local)                  ! the last operator variant's records travel
    call xee_check(xee_ensemble_destroy(ens), "ensemble_destroy")
end if
call system_clock(clk1)

if (c_associated(comm)) then
    call xee_check(xee_gather_scalars(comm, mine, per, everyone), "gather_scalars")
    call xee_check(xee_comm_destroy(comm), "comm_destroy")
else if (gather_by_files) then                       ! rehearsal on one GPU: the same records, through the file system
    write (rec_file, '(A,A,I0)') trim(comm_file), ".rec", rank
    open (unit=17, file=trim(rec_file)//".tmp", access="stream", form="unformatted", status="replace")
    write (17) mine
    close (17)
    call execute_command_line("mv -f '" // trim(rec_file) // ".tmp' '" // trim(rec_file) // "'")
    if (rank == 0) then
        do k = 0, world - 1
            write (rec_file, '(A,A,I0)') trim(comm_file), ".rec", k
            file_exists = .false.
            do i = 1, int(comm_timeout * 10.0d0) + 1
                inquire(file=trim(rec_file), exist=file_exists)
                if (file_exists) exit
                call execute_command_line("sleep 0.1")
            end do
            if (.not. file_exists) then
                call error_msg("GATHER", 1, "no records from rank: " // trim(rec_file)); error stop 1
            end if
            open (unit=17, file=trim(rec_file), access="stream", form="unformatted", status="old")
            read (17) everyone(k*per+1:(k+1)*per)
            close (17)
            call execute_command_line("rm -f '" // trim(rec_file) // "'")
        end do
        call execute_command_line("rm -f '" // trim(comm_file) // "'")
    end if
else
    everyone(1:per) = mine(1:per)
end if

if (rank == 0) then
    call get_environment_variable("XEE_SUMMARY_FILE", buffer, status=ios)
    if (ios /= 0 .or. len_trim(buffer) == 0) buffer = trim(case_dir(1)) // "/../ensemble_summary.txt"
    open (unit=15, file=trim(buffer), action="write", status="replace")
    write (15,'(A,ES16.9)') " Time elapsed (sec) : ", real(clk1 - clk0) / real(clk_rate)
    write (15,'(A)') " # case iters err_bits err_now ratio sum_Q sum_Qeta efficiency"
    do k = 1, per * world
        if (everyone(k)%case_id < 0) cycle
        write (15,'(1X,I0,1X,I0,1X,I0,5(1X,ES16.9))') everyone(k)%case_id, everyone(k)%iters, everyone(k)%err_bits, &
        &     everyone(k)%err_now, everyone(k)%ratio, everyone(k)%sum_Q, everyone(k)%sum_Qeta, everyone(k)%efficiency
    end do
    close (15)
    print *, "Diagnose complete."
end if

contains

subroutine env_int(n1, n2, n3, default, v)
    character(*), intent(in) :: n1, n2, n3
    integer, intent(in) :: default
    integer, intent(out) :: v
    integer :: status, io
    v = default
    call get_environment_variable(n1, val, status=status)
    if (status /= 0 .or. len_trim(val) == 0) call get_environment_variable(n2, val, status=status)
    if (status /= 0 .or. len_trim(val) == 0) call get_environment_variable(n3, val, status=status)
    if (status == 0 .and. len_trim(val) > 0) then
        read (val, *, iostat=io) v
        if (io /= 0) v = default
    end if
end subroutine

! seconds since 2000-01-01 (local time; the ranks of one node share a clock)
integer(8) function wall_seconds()
    integer :: v(8), y, mth, d
    integer(8) :: days
    call date_and_time(values=v)
    y = v(1); mth = v(2); d = v(3)
    if (mth <= 2) then
        y = y - 1; mth = mth + 12
    end if
    days = 365_8 * y + y / 4 - y / 100 + y / 400 + (153 * (mth - 3) + 2) / 5 + d - 730426_8
    wall_seconds = days * 86400_8 + v(5) * 3600_8 + v(6) * 60_8 + v(7)
end function

! the first of the named environment variables that is set and not blank ('' if none)
subroutine first_env(names, value)
    character(*), intent(in) :: names(:)
    character(*), intent(out) :: value
    integer :: status, n
    value = ''
    do n = 1, size(names)
        call get_environment_variable(trim(names(n)), value, status=status)
        if (status == 0 .and. len_trim(value) > 0) return
        value = ''
    end do
end subroutine

logical function env_set(name)
    character(*), intent(in) :: name
    integer :: status
    call get_environment_variable(name, val, status=status)
    env_set = status == 0 .and. len_trim(val) > 0
end function

end program xee_ensemble
