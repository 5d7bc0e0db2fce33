﻿!mod$ v1 sum:9aee14782192edfc
!need$ 99145601f71fb607 n m_base_case
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ f4f3b1cdb42159bf n m_mesh
!need$ 85f841a7a38b0974 n m_solver
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ 939e7b51cda90705 n m_allocator
!need$ f39a1ef65bd4689d n m_base_backend
module m_case_generic
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_allocator,only:allocator_t
use m_base_backend,only:base_backend_t
use m_base_case,only:base_case_t
use m_common,only:dp
use m_common,only:vert
use m_field,only:field_t
use m_mesh,only:mesh_t
use m_solver,only:init
type,extends(base_case_t)::case_generic_t
contains
procedure::define_bc=>define_bc_generic
procedure::initial_conditions=>initial_conditions_generic
procedure::forcings=>forcings_generic
procedure::apply_bc=>apply_bc_generic
procedure::postprocess=>postprocess_generic
end type
interface case_generic_t
procedure::case_generic_init
end interface
contains
function case_generic_init(backend,mesh,host_allocator) result(flow_case)
class(base_backend_t),intent(inout),target::backend
type(mesh_t),intent(inout),target::mesh
type(allocator_t),intent(inout),target::host_allocator
type(case_generic_t)::flow_case
end
subroutine define_bc_generic(self)
class(case_generic_t)::self
end
subroutine initial_conditions_generic(self)
class(case_generic_t)::self
end
subroutine forcings_generic(self,du,dv,dw,iter)
class(case_generic_t)::self
class(field_t),intent(inout)::du
class(field_t),intent(inout)::dv
class(field_t),intent(inout)::dw
integer(4),intent(in)::iter
end
subroutine apply_bc_generic(self,u,v,w)
class(case_generic_t)::self
class(field_t),intent(inout)::u
class(field_t),intent(inout)::v
class(field_t),intent(inout)::w
end
subroutine postprocess_generic(self,iter,t)
class(case_generic_t)::self
integer(4),intent(in)::iter
real(8),intent(in)::t
end
end
