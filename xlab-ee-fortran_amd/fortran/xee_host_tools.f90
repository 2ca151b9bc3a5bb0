! xee_host_tools.f90 -- the host-side surface the Fortran driver keeps (north star: "existing
! config and file I/O surface"): same module and procedure names as the reference's support
! library so that driver code written against it compiles unchanged, re-implemented from the
! formats they define (SURVEY.md Appendix A):
!
!   module read_input_tools : read_input, split_line      (xtt-lib-fortran/read_input_tools.f90:7-62)
!   module message_tools    : error_msg, system_msg       (xtt-lib-fortran/message_tools.f90:6-20)
!   module field_tools      : read_2Dfield, write_2Dfield, write_1Dfield, read_3Dfield,
!                             write_3Dfield              (xtt-lib-fortran/field_tools.f90:5-123)
!   module constants        : g0 ... PI, print_constants  (xtt-lib-fortran/constants.f90:4-26)
!
! Wire formats: a field file is one raw little-endian float32 record, i (r) fastest, no record
! markers -- written here with STREAM access, which produces the same bytes as the reference's
! direct-access record of recl = 4*nx*ny.  A config line ends at "//", blank lines are skipped,
! at most 256 characters are kept.

module constants
implicit none
real(4), parameter :: g0=9.8, theta0=298.0, Rd=287.0, Cv=5.0/2.0*Rd, Cp=Cv+Rd, kappa=Rd/Cp, &
&                     h0=Cp*theta0/g0, p0=101300.0
real(4), parameter :: PI = acos(-1.0), PII = 2*PI
contains
subroutine print_constants()
    print *, "Constants in xtt-lib/constants.f90 : "
    print *, "Rd      (J/K/kg) : ", Rd
    print *, "Cv      (J/K/kg) : ", Cv
    print *, "Cp      (J/K/kg) : ", Cp
    print *, "kappa            : ", kappa
    print *, "g0      (m/s^2)  : ", g0
    print *, "p0      (pa)     : ", p0
    print *, "theta0  (K)      : ", theta0
    print *, "h0      (m)      : ", h0
end subroutine
end module constants


module message_tools
implicit none
contains
subroutine error_msg(err_type, err_code, err_msg)
    character(*), intent(in) :: err_type, err_msg
    integer, intent(in) :: err_code
    write (*, "(A,A,A,I3,A,A)") "ERROR: [", trim(err_type), ",", err_code, "] : ", trim(err_msg)
end subroutine
subroutine system_msg(sys_type, sys_msg)
    character(*), intent(in) :: sys_type, sys_msg
    write (*, "(A,A,A,A)") "[", trim(sys_type), "] : ", trim(sys_msg)
end subroutine
end module message_tools


module read_input_tools
implicit none
contains

! next non-empty line of unit fd with everything from "//" on removed (and a trailing CR, so that
! CRLF files such as the shipped test/test1/diag.txt work even without a comment on the line)
subroutine read_input(fd, line)
    integer, intent(in) :: fd
    character(256), intent(out) :: line
    character(256) :: raw
    integer :: cut, n

    do
        read (fd, '(a)') raw
        cut = index(raw, '//')
        if (cut > 0) raw(cut:) = ' '
        n = len_trim(raw)
        if (n > 0) then
            if (raw(n:n) == achar(13)) raw(n:n) = ' '
        end if
        if (len_trim(raw) > 0) exit
    end do
    line = raw
end subroutine

! pops the text before the first delimiter off `line`; 1 when no delimiter was left, else 0
integer function split_line(line, output, delimiter)
    character(*), intent(inout) :: line, output
    character(*), intent(in)    :: delimiter
    integer :: k

    k = index(line, delimiter)
    if (k == 0) then
        output = line
        line = ''
        split_line = 1
    else
        output = line(1:k-1)
        line = line(k+1:)
        split_line = 0
    end if
end function
end module read_input_tools


module field_tools
implicit none
private
public :: read_2Dfield, write_2Dfield, write_1Dfield, read_3Dfield, write_3Dfield
contains

subroutine put_raw(fd, filename, f, n)
    integer, intent(in) :: fd, n
    character(len=*), intent(in) :: filename
    real(4), intent(in) :: f(n)
    integer :: ios
    open (fd, file=filename, access="stream", form="unformatted", status="replace", iostat=ios)
    if (ios /= 0) then
        print *, "Writing field error. File name: ", trim(filename)
        return
    end if
    write (fd, iostat=ios) f
    close (fd)
    if (ios /= 0) print *, "Writing field error. File name: ", trim(filename)
end subroutine

subroutine get_raw(fd, filename, f, n)
    integer, intent(in) :: fd, n
    character(len=*), intent(in) :: filename
    real(4), intent(inout) :: f(n)
    integer :: ios
    open (fd, file=filename, access="stream", form="unformatted", status="old", action="read", iostat=ios)
    if (ios /= 0) then
        print *, "Reading field error. File name: ", trim(filename)
        print *, "Error number: ", ios
        return
    end if
    read (fd, iostat=ios) f
    close (fd)
    if (ios /= 0) print *, "Reading field error. File name: ", trim(filename)
end subroutine

subroutine write_1Dfield(fd, filename, f, nx)
    character(len=*) :: filename
    integer, intent(in) :: fd, nx
    real(4), intent(in) :: f(nx)
    call put_raw(fd, filename, f, nx)
end subroutine

subroutine read_2Dfield(fd, filename, f, nx, ny)
    character(len=*) :: filename
    integer, intent(in) :: fd, nx, ny
    real(4), intent(inout) :: f(nx,ny)
    call get_raw(fd, filename, f, nx*ny)
end subroutine

subroutine write_2Dfield(fd, filename, f, nx, ny)
    character(len=*) :: filename
    integer, intent(in) :: fd, nx, ny
    real(4), intent(in) :: f(nx,ny)
    call put_raw(fd, filename, f, nx*ny)
end subroutine

subroutine read_3Dfield(fd, filename, f, nx, ny, nz)
    character(len=*) :: filename
    integer, intent(in) :: fd, nx, ny, nz
    real(4), intent(inout) :: f(nx,ny,nz)
    call get_raw(fd, filename, f, nx*ny*nz)
end subroutine

subroutine write_3Dfield(fd, filename, f, nx, ny, nz)
    character(len=*) :: filename
    integer, intent(in) :: fd, nx, ny, nz
    real(4), intent(in) :: f(nx,ny,nz)
    call put_raw(fd, filename, f, nx*ny*nz)
end subroutine
end module field_tools
