! elliptic_tools.f90 -- drop-in replacement for the reference module of the same name
! (xtt-lib-fortran/elliptic_tools.f90).  Same module name, same public procedures, same argument
! lists and intents, so a driver that says `use elliptic_tools` (src/diagnose/main.f90:2) links
! against this file unchanged; the work happens in libxee (hand-written HIP kernels for gfx950)
! through the ISO_C_BINDING interfaces below (include/xee.h).
!
!   cal_coe        <- elliptic_tools.f90:8-60      -> xee_cal_coe
!   do_elliptic    <- elliptic_tools.f90:64-90     -> xee_do_elliptic
!   solve_elliptic <- elliptic_tools.f90:93-265    -> xee_solve_elliptic
!   judge_error    <- elliptic_tools.f90:333-358   -> xee_judge_error
!
! Error convention: libxee never exits.  Where the reference STOPs (both criteria non-positive,
! :126-129) this shim STOPs; any other non-zero status (no GPU, HIP failure) is fatal with the
! library's message -- there is no CPU fallback.
module elliptic_tools
use, intrinsic :: iso_c_binding
implicit none
private
public :: cal_coe, do_elliptic, solve_elliptic, judge_error, err_over_max_iteration, err_explode

integer, parameter :: err_over_max_iteration = ISHFT(1,0), &
    &                 err_explode = ISHFT(1,1)

integer(c_int), parameter :: XEE_OK = 0, XEE_ERR_INVALID_CRITERIA = 1

interface
    function xee_cal_coe(a, b, c, coe, dx, dy, nx, ny, err) bind(C, name="xee_cal_coe") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)     :: a(*), b(*), c(*)
        real(c_float), intent(inout)  :: coe(*)
        real(c_float), value          :: dx, dy
        integer(c_int), value         :: nx, ny
        integer(c_int), intent(inout) :: err
        integer(c_int)                :: st
    end function

    function xee_do_elliptic(psi, coe, outdat, nx, ny, err) bind(C, name="xee_do_elliptic") result(st)
        import :: c_int, c_float
        real(c_float), intent(in)     :: psi(*), coe(*)
        real(c_float), intent(inout)  :: outdat(*)
        integer(c_int), value         :: nx, ny
        integer(c_int), intent(inout) :: err
        integer(c_int)                :: st
    end function

    function xee_solve_elliptic(max_iter, check_step, converge_time, lost_rate, strategy_r1,     &
        &                       strategy_r2, alpha, dat, coe, f, workspace, nx, ny, err, debug)  &
        &                       bind(C, name="xee_solve_elliptic") result(st)
        import :: c_int, c_float
        integer(c_int), intent(inout) :: max_iter, err
        integer(c_int), value         :: check_step, converge_time, lost_rate, nx, ny, debug
        real(c_float), intent(inout)  :: strategy_r1, strategy_r2
        real(c_float), value          :: alpha
        real(c_float), intent(inout)  :: dat(*), workspace(*)
        real(c_float), intent(in)     :: coe(*), f(*)
        integer(c_int)                :: st
    end function

    subroutine xee_judge_error(err) bind(C, name="xee_judge_error")
        import :: c_int
        integer(c_int), value :: err
    end subroutine

    function xee_last_error() bind(C, name="xee_last_error") result(p)
        import :: c_ptr
        type(c_ptr) :: p
    end function
end interface

contains

subroutine xee_fatal(where, st)
implicit none
character(*), intent(in) :: where
integer(c_int), intent(in) :: st
character(kind=c_char), pointer :: msg(:)
type(c_ptr) :: p
integer :: n

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

subroutine cal_coe(a, b, c, workspace, dx, dy, nx, ny, err)
implicit none
real(4), intent(in) :: a(nx-1, ny-2), b(nx-1, ny-1), c(nx-2, ny-1), &
&                      dx, dy
real(4), intent(inout) :: workspace(9, nx, ny)
integer, intent(in)    :: nx, ny
integer, intent(inout) :: err
integer(c_int) :: st

st = xee_cal_coe(a, b, c, workspace, dx, dy, nx, ny, err)
if (st /= XEE_OK) call xee_fatal("cal_coe", st)
end subroutine

subroutine do_elliptic(psi, coe, outdat, nx, ny, err)
implicit none
real(4), intent(in)    :: psi(nx, ny), coe(9, nx,ny)
real(4), intent(inout) :: outdat(nx,ny)
integer, intent(in)    :: nx, ny
integer, intent(inout) :: err
integer(c_int) :: st

st = xee_do_elliptic(psi, coe, outdat, nx, ny, err)
if (st /= XEE_OK) call xee_fatal("do_elliptic", st)
end subroutine

subroutine solve_elliptic(max_iter, check_step, converge_time, lost_rate, &
&                         strategy_r1, strategy_r2, alpha, dat, coe, f, &
&                         workspace, nx, ny, err, debug)
implicit none
real(4), intent(inout), target :: dat(nx,ny), workspace(nx, ny)
real(4), intent(inout) :: strategy_r1, strategy_r2
real(4), intent(in)    :: coe(9, nx, ny), f(nx, ny), alpha
integer, intent(in)    :: nx, ny, check_step, converge_time, lost_rate, debug
integer, intent(inout) :: max_iter, err
integer(c_int) :: st
integer :: max_iter_in

max_iter_in = max_iter
! 256 = XEE_DEBUG_NO_JUDGE: the judge_error text is printed below through Fortran I/O
st = xee_solve_elliptic(max_iter, check_step, converge_time, lost_rate, strategy_r1, strategy_r2, &
&                       alpha, dat, coe, f, workspace, nx, ny, err, ior(debug, 256))
if (st == XEE_ERR_INVALID_CRITERIA) then
    ! the library has printed the reference's message (elliptic_tools.f90:127); the reference STOPs here
    stop
else if (st /= XEE_OK) then
    call xee_fatal("solve_elliptic", st)
end if
if (max_iter_in > 0) call judge_error(err)      ! elliptic_tools.f90:254
end subroutine

subroutine judge_error(err)
implicit none
integer, intent(in) :: err
logical :: known_err

! same texts as xtt-lib-fortran/elliptic_tools.f90:340-357, printed through Fortran I/O so that
! they interleave correctly with the driver's own print statements
known_err = .false.
if (err == 0) then
    print *, "Elliptic Tools: Iteration success."
    known_err = .true.
end if
if (IAND(err, err_over_max_iteration) /= 0) then
    print *, "Elliptic Tools: [Error] Max iteration reached."
    known_err = .true.
end if
if (IAND(err, err_explode) /= 0) then
    print *, "Elliptic Tools: [Error] Iteration explodes."
    known_err = .true.
end if
if (known_err .eqv. .false.) then
    print *, "Elliptic Tools: Unknown error code ", err
end if
end subroutine

end module elliptic_tools
