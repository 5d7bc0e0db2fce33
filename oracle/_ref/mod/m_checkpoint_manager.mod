﻿!mod$ v1 sum:01bb74536fd4092c
!need$ dc0e186b83d0ae94 n m_checkpoint_state
!need$ 078593e6885e6bc7 n m_stats
!need$ a1f26d8334a87c7c n m_io_field_utils
!need$ 0df96a70750958ab n mpi
!need$ 7f5e804034ee5163 n m_config
!need$ fadd42cafe0c8e6b n m_io_session
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ 85f841a7a38b0974 n m_solver
module m_checkpoint_manager
use mpi,only:mpi_comm_world
use mpi,only:mpi_comm_rank
use mpi,only:mpi_abort
use m_common,only:dp
use m_common,only:i8
use m_common,only:dir_x
use m_common,only:get_argument
use m_field,only:field_t
use m_solver,only:solver_t
use m_io_session,only:reader_session_t
use m_io_session,only:writer_session_t
use m_config,only:checkpoint_config_t
use m_checkpoint_state,only:checkpoint_state_t
use m_stats,only:stats_manager_t
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
private::mpi_abort
private::dp
private::i8
private::dir_x
private::get_argument
private::field_t
private::solver_t
private::reader_session_t
private::writer_session_t
private::checkpoint_config_t
private::checkpoint_state_t
private::stats_manager_t
private::field_buffer_map_t
private::field_ptr_t
private::setup_field_arrays
private::cleanup_field_arrays
private::stride_data_to_buffer
private::get_output_dimensions
private::prepare_field_buffers
private::cleanup_field_buffers
private::write_single_field_to_buffer
type,private::raw_old_field_buffer_t
real(8),allocatable::data(:,:,:)
end type
type::checkpoint_manager_t
type(checkpoint_config_t)::config
integer(4)::last_checkpoint_step=-1_4
integer(4)::full_resolution(1_8:3_8)=[INTEGER(4)::1_4,1_4,1_4]
type(field_buffer_map_t),allocatable::field_buffers(:)
integer(8)::last_shape_dims(1_8:3_8)=[INTEGER(8)::0_8,0_8,0_8]
integer(4)::last_stride_factors(1_8:3_8)=[INTEGER(4)::0_4,0_4,0_4]
integer(8)::last_output_shape(1_8:3_8)=[INTEGER(8)::0_8,0_8,0_8]
contains
procedure::init
procedure::handle_restart
procedure::handle_checkpoint_step
procedure::restore_state
procedure::is_restart
procedure::finalise
procedure,private::write_checkpoint
procedure,private::restart_checkpoint
procedure,private::write_fields
procedure,private::cleanup_output_buffers
end type
private::init
private::configure_output
private::is_restart
private::handle_restart
private::handle_checkpoint_step
private::write_checkpoint
private::restore_state
private::restart_checkpoint
private::write_fields
private::cleanup_output_buffers
private::finalise
contains
subroutine init(self,comm)
class(checkpoint_manager_t),intent(inout)::self
integer(4),intent(in)::comm
end
subroutine configure_output(self,comm)
class(checkpoint_manager_t),intent(inout)::self
integer(4),intent(in)::comm
end
function is_restart(self) result(restart)
class(checkpoint_manager_t),intent(in)::self
logical(4)::restart
end
subroutine handle_restart(self,solver,comm,stats_mgr)
class(checkpoint_manager_t),intent(inout)::self
class(solver_t),intent(inout)::solver
integer(4),intent(in),optional::comm
type(stats_manager_t),intent(inout),optional::stats_mgr
end
subroutine handle_checkpoint_step(self,solver,timestep,comm,stats_mgr,checkpoint_state)
class(checkpoint_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::timestep
integer(4),intent(in),optional::comm
type(stats_manager_t),intent(inout),optional::stats_mgr
class(checkpoint_state_t),intent(inout),optional::checkpoint_state
end
subroutine write_checkpoint(self,solver,timestep,comm,stats_mgr,checkpoint_state)
class(checkpoint_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::timestep
integer(4),intent(in)::comm
type(stats_manager_t),intent(inout),optional::stats_mgr
class(checkpoint_state_t),intent(inout),optional::checkpoint_state
end
subroutine restore_state(self,checkpoint_state,comm)
class(checkpoint_manager_t),intent(inout)::self
class(checkpoint_state_t),intent(inout)::checkpoint_state
integer(4),intent(in)::comm
end
subroutine restart_checkpoint(self,solver,filename,timestep,restart_time,comm,stats_mgr)
class(checkpoint_manager_t),intent(inout)::self
class(solver_t),intent(inout)::solver
character(*,1),intent(in)::filename
integer(4),intent(out)::timestep
real(8),intent(out)::restart_time
integer(4),intent(in)::comm
type(stats_manager_t),intent(inout),optional::stats_mgr
end
subroutine write_fields(self,field_names,host_fields,solver,writer_session,data_loc)
class(checkpoint_manager_t),intent(inout)::self
character(*,1),intent(in)::field_names(:)
class(field_ptr_t),intent(in),target::host_fields(:)
class(solver_t),intent(in)::solver
type(writer_session_t),intent(inout)::writer_session
integer(4),intent(in)::data_loc
end
subroutine cleanup_output_buffers(self)
class(checkpoint_manager_t),intent(inout)::self
end
subroutine finalise(self)
class(checkpoint_manager_t),intent(inout)::self
end
end
