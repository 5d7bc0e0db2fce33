﻿!mod$ v1 sum:99fc4d48f3e44c01
!need$ fadd42cafe0c8e6b n m_io_session
!need$ a1f26d8334a87c7c n m_io_field_utils
!need$ 7f5e804034ee5163 n m_config
!need$ 0df96a70750958ab n mpi
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ 85f841a7a38b0974 n m_solver
module m_snapshot_manager
use mpi,only:mpi_comm_world
use mpi,only:mpi_comm_rank
use m_common,only:dp
use m_common,only:i8
use m_common,only:dir_c
use m_common,only:vert
use m_common,only:get_argument
use m_field,only:field_t
use m_solver,only:solver_t
use m_io_session,only:writer_session_t
use m_config,only:checkpoint_config_t
use m_config,only:has_output_field
use m_io_field_utils,only:field_buffer_map_t
use m_io_field_utils,only:field_ptr_t
use m_io_field_utils,only:setup_field_arrays
use m_io_field_utils,only:cleanup_field_arrays
use m_io_field_utils,only:stride_data_to_buffer
use m_io_field_utils,only:get_output_dimensions
use m_io_field_utils,only:prepare_field_buffers
use m_io_field_utils,only:cleanup_field_buffers
use m_io_field_utils,only:write_single_field_to_buffer
private::mpi_comm_world
private::mpi_comm_rank
private::dp
private::i8
private::dir_c
private::vert
private::get_argument
private::field_t
private::solver_t
private::writer_session_t
private::checkpoint_config_t
private::has_output_field
private::field_buffer_map_t
private::field_ptr_t
private::setup_field_arrays
private::cleanup_field_arrays
private::stride_data_to_buffer
private::get_output_dimensions
private::prepare_field_buffers
private::cleanup_field_buffers
private::write_single_field_to_buffer
type::snapshot_manager_t
type(checkpoint_config_t)::config
integer(4)::output_stride(1_8:3_8)=[INTEGER(4)::1_4,1_4,1_4]
type(field_buffer_map_t),allocatable::field_buffers(:)
integer(8)::last_shape_dims(1_8:3_8)=[INTEGER(8)::0_8,0_8,0_8]
integer(4)::last_stride_factors(1_8:3_8)=[INTEGER(4)::0_4,0_4,0_4]
integer(8)::last_output_shape(1_8:3_8)=[INTEGER(8)::0_8,0_8,0_8]
character(:,1),allocatable::vtk_xml
logical(4)::is_snapshot_file_open=.false._4
type(writer_session_t)::snapshot_writer
logical(4)::convert_to_sp=.false._4
contains
procedure::init
procedure::handle_snapshot_step
procedure::finalise
procedure,private::write_snapshot
procedure,private::write_fields
procedure,private::cleanup_output_buffers
procedure,private::generate_vtk_xml
procedure,private::open_snapshot_file
procedure,private::close_snapshot_file
end type
private::init
private::configure_output
private::handle_snapshot_step
private::write_snapshot
private::get_snapshot_fields
private::generate_vtk_xml
private::write_fields
private::cleanup_output_buffers
private::finalise
private::open_snapshot_file
private::close_snapshot_file
private::join_output_fields
contains
subroutine init(self,comm)
class(snapshot_manager_t),intent(inout)::self
integer(4),intent(in)::comm
end
subroutine configure_output(self,comm)
class(snapshot_manager_t),intent(inout)::self
integer(4),intent(in)::comm
end
subroutine handle_snapshot_step(self,solver,timestep,comm)
class(snapshot_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::timestep
integer(4),intent(in),optional::comm
end
subroutine write_snapshot(self,solver,timestep,comm)
class(snapshot_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::timestep
integer(4),intent(in)::comm
end
function get_snapshot_fields(config,nspecies) result(names)
type(checkpoint_config_t),intent(in)::config
integer(4),intent(in)::nspecies
character(32_4,1),allocatable::names(:)
end
subroutine generate_vtk_xml(self,dims,fields,origin,spacing)
class(snapshot_manager_t),intent(inout)::self
integer(8),intent(in)::dims(1_8:3_8)
character(*,1),intent(in)::fields(:)
real(8),intent(in)::origin(1_8:3_8)
real(8),intent(in)::spacing(1_8:3_8)
end
subroutine write_fields(self,field_names,host_fields,solver,writer_session,data_loc)
class(snapshot_manager_t),intent(inout)::self
character(*,1),intent(in)::field_names(:)
class(field_ptr_t),intent(in),target::host_fields(:)
class(solver_t),intent(in)::solver
type(writer_session_t),intent(inout)::writer_session
integer(4),intent(in)::data_loc
end
subroutine cleanup_output_buffers(self)
class(snapshot_manager_t),intent(inout)::self
end
subroutine finalise(self)
class(snapshot_manager_t),intent(inout)::self
end
subroutine open_snapshot_file(self,filename,comm)
class(snapshot_manager_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::comm
end
subroutine close_snapshot_file(self)
class(snapshot_manager_t),intent(inout)::self
end
function join_output_fields(fields) result(str)
character(32_4,1),intent(in)::fields(1_8:10_8)
character(256_4,1)::str
end
end
