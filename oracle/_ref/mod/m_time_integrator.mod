﻿!mod$ v1 sum:bca18982b31d732a
!need$ f39a1ef65bd4689d n m_base_backend
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ 939e7b51cda90705 n m_allocator
module m_time_integrator
use m_allocator,only:allocator_t
use m_base_backend,only:base_backend_t
use m_common,only:dp
use m_common,only:dir_x
use m_field,only:field_t
use m_field,only:flist_t
type::time_intg_t
integer(4)::method
integer(4)::istep
integer(4)::istage
integer(4)::order
integer(4)::nstep
integer(4)::nstage
integer(4)::nvars
integer(4)::nolds
real(8)::coeffs(1_8:4_8,1_8:4_8)
real(8)::rk_b(1_8:4_8,1_8:4_8)
real(8)::rk_a(1_8:3_8,1_8:3_8,1_8:4_8)
real(8)::gdt
character(3_4,1)::sname
type(flist_t),allocatable::olds(:,:)
class(base_backend_t),pointer::backend
class(allocator_t),pointer::allocator
procedure(stepper_func),pointer::step
contains
procedure::finalize
procedure::runge_kutta
procedure::adams_bashforth
end type
intrinsic::null
abstract interface
subroutine stepper_func(self,curr,deriv,dt)
import::flist_t
import::time_intg_t
class(time_intg_t),intent(inout)::self
type(flist_t),intent(inout)::curr(:)
type(flist_t),intent(in)::deriv(:)
real(8),intent(in)::dt
end
end interface
private::runge_kutta
private::adams_bashforth
interface time_intg_t
procedure::init
end interface
contains
subroutine finalize(self)
class(time_intg_t),intent(inout)::self
end
function init(backend,allocator,method,nvars)
class(base_backend_t),pointer::backend
class(allocator_t),pointer::allocator
character(3_4,1),intent(in)::method
integer(4),intent(in)::nvars
type(time_intg_t)::init
end
subroutine runge_kutta(self,curr,deriv,dt)
class(time_intg_t),intent(inout)::self
type(flist_t),intent(inout)::curr(:)
type(flist_t),intent(in)::deriv(:)
real(8),intent(in)::dt
end
subroutine adams_bashforth(self,curr,deriv,dt)
class(time_intg_t),intent(inout)::self
type(flist_t),intent(inout)::curr(:)
type(flist_t),intent(in)::deriv(:)
real(8),intent(in)::dt
end
subroutine rotate(sol,n)
type(flist_t),intent(inout)::sol(:)
integer(4),intent(in)::n
end
end
