﻿!mod$ v1 sum:19f0b2ec919eb9c3
!need$ 01bb74536fd4092c n m_checkpoint_manager
!need$ dc0e186b83d0ae94 n m_checkpoint_state
!need$ 99fc4d48f3e44c01 n m_snapshot_manager
!need$ 078593e6885e6bc7 n m_stats
!need$ 85f841a7a38b0974 n m_solver
!need$ 0df96a70750958ab n mpi
module m_io_manager
use mpi,only:mpi_comm_world
use m_checkpoint_manager,only:checkpoint_manager_t
use m_checkpoint_state,only:checkpoint_state_t
use m_snapshot_manager,only:snapshot_manager_t
use m_stats,only:stats_manager_t
use m_solver,only:solver_t
private::mpi_comm_world
private::checkpoint_manager_t
private::checkpoint_state_t
private::snapshot_manager_t
private::stats_manager_t
private::solver_t
type::io_manager_t
type(checkpoint_manager_t)::checkpoint_mgr
type(snapshot_manager_t)::snapshot_mgr
type(stats_manager_t)::stats_mgr
class(checkpoint_state_t),pointer::additional_checkpoint_state=>NULL()
contains
procedure::init=>io_init
procedure::handle_restart=>io_handle_restart
procedure::handle_io_step=>io_handle_step
procedure::update_stats=>io_update_stats
procedure::register_checkpoint_state
procedure::unregister_checkpoint_state
procedure::finalise=>io_finalise
procedure::is_restart=>io_is_restart
end type
intrinsic::null
private::null
private::io_init
private::io_handle_restart
private::io_update_stats
private::register_checkpoint_state
private::unregister_checkpoint_state
private::io_handle_step
private::io_is_restart
private::io_finalise
contains
subroutine io_init(self,solver,comm)
class(io_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::comm
end
subroutine io_handle_restart(self,solver,comm)
class(io_manager_t),intent(inout)::self
class(solver_t),intent(inout)::solver
integer(4),intent(in),optional::comm
end
subroutine io_update_stats(self,solver,iter)
class(io_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::iter
end
subroutine register_checkpoint_state(self,checkpoint_state,comm)
class(io_manager_t),intent(inout)::self
class(checkpoint_state_t),intent(inout),target::checkpoint_state
integer(4),intent(in),optional::comm
end
subroutine unregister_checkpoint_state(self)
class(io_manager_t),intent(inout)::self
end
subroutine io_handle_step(self,solver,timestep,comm)
class(io_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::timestep
integer(4),intent(in),optional::comm
end
function io_is_restart(self) result(is_restart)
class(io_manager_t),intent(in)::self
logical(4)::is_restart
end
subroutine io_finalise(self)
class(io_manager_t),intent(inout)::self
end
end
