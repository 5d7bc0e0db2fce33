﻿!mod$ v1 sum:e23ea590ed584e12
!need$ 99145601f71fb607 n m_base_case
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ f4f3b1cdb42159bf n m_mesh
!need$ 85f841a7a38b0974 n m_solver
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ 939e7b51cda90705 n m_allocator
!need$ f39a1ef65bd4689d n m_base_backend
module m_case_tgv
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_allocator,only:allocator_t
use m_base_backend,only:base_backend_t
use m_base_case,only:base_case_t
use m_common,only:dp
use m_common,only:vert
use m_common,only:dir_c
use m_field,only:field_t
use m_mesh,only:mesh_t
use m_solver,only:init
type,extends(base_case_t)::case_tgv_t
contains
procedure::define_bc=>define_bc_tgv
procedure::initial_conditions=>initial_conditions_tgv
procedure::forcings=>forcings_tgv
procedure::apply_bc=>apply_bc_tgv
procedure::postprocess=>postprocess_tgv
end type
interface case_tgv_t
procedure::case_tgv_init
end interface
contains
function case_tgv_init(backend,mesh,host_allocator) result(flow_case)
class(base_backend_t),intent(inout),target::backend
type(mesh_t),intent(inout),target::mesh
type(allocator_t),intent(inout),target::host_allocator
type(case_tgv_t)::flow_case
end
subroutine initial_conditions_tgv(self)
class(case_tgv_t)::self
end
pure function u_func(coords) result(r)
real(8),intent(in)::coords(1_8:3_8)
real(8)::r
end
pure function v_func(coords) result(r)
real(8),intent(in)::coords(1_8:3_8)
real(8)::r
end
subroutine define_bc_tgv(self)
class(case_tgv_t)::self
end
subroutine forcings_tgv(self,du,dv,dw,iter)
class(case_tgv_t)::self
class(field_t),intent(inout)::du
class(field_t),intent(inout)::dv
class(field_t),intent(inout)::dw
integer(4),intent(in)::iter
end
subroutine apply_bc_tgv(self,u,v,w)
class(case_tgv_t)::self
class(field_t),intent(inout)::u
class(field_t),intent(inout)::v
class(field_t),intent(inout)::w
end
subroutine postprocess_tgv(self,iter,t)
class(case_tgv_t)::self
integer(4),intent(in)::iter
real(8),intent(in)::t
end
end
