﻿!mod$ v1 sum:99145601f71fb607
!need$ 2785dcb69f8c0771 n m_monitoring
!need$ f4f3b1cdb42159bf n m_mesh
!need$ e03d3be8503abc60 n m_postprocess
!need$ 19f0b2ec919eb9c3 n m_io_manager
!need$ 0df96a70750958ab n mpi
!need$ 939e7b51cda90705 n m_allocator
!need$ f39a1ef65bd4689d n m_base_backend
!need$ 85f841a7a38b0974 n m_solver
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ 7f5e804034ee5163 n m_config
module m_base_case
use m_allocator,only:allocator_t
use m_base_backend,only:base_backend_t
use m_common,only:dp
use m_common,only:dir_x
use m_common,only:dir_c
use m_common,only:vert
use m_common,only:mpi_x3d2_dp
use m_monitoring,only:monitoring_t
use m_field,only:field_t
use m_field,only:flist_t
use m_mesh,only:mesh_t
use m_solver,only:solver_t
use m_solver,only:init
use m_postprocess,only:compute_derived_fields
use m_postprocess,only:compute_pressure_vert
use m_config,only:has_output_field
use m_io_manager,only:io_manager_t
use mpi,only:mpi_comm_world
use mpi,only:mpi_wtime
use mpi,only:mpi_reduce
use mpi,only:mpi_max
type,abstract::base_case_t
class(solver_t),allocatable::solver
type(io_manager_t)::io_mgr
type(monitoring_t)::monitoring
class(field_t),pointer::bc_start_u_x=>NULL()
class(field_t),pointer::bc_start_v_x=>NULL()
class(field_t),pointer::bc_start_w_x=>NULL()
class(field_t),pointer::bc_end_u_x=>NULL()
class(field_t),pointer::bc_end_v_x=>NULL()
class(field_t),pointer::bc_end_w_x=>NULL()
class(field_t),pointer::bc_start_u_y=>NULL()
class(field_t),pointer::bc_start_v_y=>NULL()
class(field_t),pointer::bc_start_w_y=>NULL()
class(field_t),pointer::bc_end_u_y=>NULL()
class(field_t),pointer::bc_end_v_y=>NULL()
class(field_t),pointer::bc_end_w_y=>NULL()
contains
procedure(define_bc),deferred::define_bc
procedure(initial_conditions),deferred::initial_conditions
procedure(forcings),deferred::forcings
procedure(apply_bc),deferred::apply_bc
procedure(postprocess),deferred::postprocess
procedure::case_init
procedure::case_finalise
procedure::set_init
procedure::run
end type
intrinsic::null
abstract interface
subroutine define_bc(self)
import::base_case_t
class(base_case_t)::self
end
end interface
abstract interface
subroutine initial_conditions(self)
import::base_case_t
class(base_case_t)::self
end
end interface
abstract interface
subroutine forcings(self,du,dv,dw,iter)
import::base_case_t
import::field_t
class(base_case_t)::self
class(field_t),intent(inout)::du
class(field_t),intent(inout)::dv
class(field_t),intent(inout)::dw
integer(4),intent(in)::iter
end
end interface
abstract interface
subroutine apply_bc(self,u,v,w)
import::base_case_t
import::field_t
class(base_case_t)::self
class(field_t),intent(inout)::u
class(field_t),intent(inout)::v
class(field_t),intent(inout)::w
end
end interface
abstract interface
subroutine postprocess(self,iter,t)
import::base_case_t
class(base_case_t)::self
integer(4),intent(in)::iter
real(8),intent(in)::t
end
end interface
contains
subroutine case_init(self,backend,mesh,host_allocator)
class(base_case_t)::self
class(base_backend_t),intent(inout),target::backend
type(mesh_t),intent(inout),target::mesh
type(allocator_t),intent(inout),target::host_allocator
end
subroutine case_finalise(self)
class(base_case_t)::self
end
subroutine set_init(self,field,field_func)
class(base_case_t)::self
class(field_t),intent(inout)::field
interface
pure function field_func(coords) result(r)
real(8),intent(in)::coords(1_8:3_8)
real(8)::r
end
end interface
end
subroutine run(self)
class(base_case_t),intent(inout)::self
end
end
