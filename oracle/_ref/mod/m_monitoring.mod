﻿!mod$ v1 sum:2785dcb69f8c0771
!need$ 4e78cfb6ee5c840a n m_scalar_series
!need$ 85f841a7a38b0974 n m_solver
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
module m_monitoring
use m_common,only:dp
use m_common,only:dir_x
use m_common,only:dir_z
use m_common,only:vert
use m_field,only:field_t
use m_scalar_series,only:scalar_series_t
use m_solver,only:solver_t
type::monitoring_t
logical(4),private::is_root=.false._4
type(scalar_series_t),private::series
contains
procedure::init
procedure::write_step
procedure::finalise
end type
contains
subroutine init(self,solver,append)
class(monitoring_t),intent(inout)::self
class(solver_t),intent(in)::solver
logical(4),intent(in),optional::append
end
subroutine write_step(self,solver,t,u,v,w)
class(monitoring_t),intent(inout)::self
class(solver_t),intent(inout)::solver
real(8),intent(in)::t
class(field_t),intent(in)::u
class(field_t),intent(in)::v
class(field_t),intent(in)::w
end
subroutine finalise(self)
class(monitoring_t),intent(inout)::self
end
end
