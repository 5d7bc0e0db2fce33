﻿!mod$ v1 sum:078593e6885e6bc7
!need$ 85f841a7a38b0974 n m_solver
!need$ fadd42cafe0c8e6b n m_io_session
!need$ 7f5e804034ee5163 n m_config
!need$ 0df96a70750958ab n mpi
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
module m_stats
use mpi,only:mpi_comm_world
use mpi,only:mpi_comm_rank
use m_common,only:dp
use m_common,only:i8
use m_common,only:dir_c
use m_common,only:dir_x
use m_common,only:vert
use m_common,only:get_argument
use m_config,only:stats_config_t
use m_field,only:field_t
use m_solver,only:solver_t
use m_io_session,only:writer_session_t
use m_io_session,only:reader_session_t
private::mpi_comm_world
private::mpi_comm_rank
private::dp
private::i8
private::dir_c
private::dir_x
private::vert
private::get_argument
private::stats_config_t
private::field_t
private::solver_t
private::writer_session_t
private::reader_session_t
type::stats_manager_t
type(stats_config_t)::config
integer(4)::sample_count=0_4
logical(4)::is_active=.false._4
real(8),allocatable::umean(:,:,:)
real(8),allocatable::vmean(:,:,:)
real(8),allocatable::wmean(:,:,:)
real(8),allocatable::uumean(:,:,:)
real(8),allocatable::vvmean(:,:,:)
real(8),allocatable::wwmean(:,:,:)
real(8),allocatable::uvmean(:,:,:)
real(8),allocatable::uwmean(:,:,:)
real(8),allocatable::vwmean(:,:,:)
real(8),allocatable::pmean(:,:,:)
integer(4)::nspecies=0_4
real(8),allocatable::phimean(:,:,:,:)
real(8),allocatable::phiphimean(:,:,:,:)
contains
procedure::init
procedure::update
procedure::write_stats
procedure::write_checkpoint
procedure::read_checkpoint
procedure::finalise
end type
private::init
private::update
private::write_stats
private::write_checkpoint
private::read_checkpoint
private::finalise
contains
pure subroutine accumulate_mean(mean,val,stat_inc)
real(8),intent(inout)::mean(:,:,:)
real(8),intent(in)::val(:,:,:)
real(8),intent(in)::stat_inc
end
subroutine init(self,solver,comm)
class(stats_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::comm
end
subroutine update(self,solver,iter)
class(stats_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::iter
end
subroutine write_stats(self,solver,timestep,comm)
class(stats_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
integer(4),intent(in)::timestep
integer(4),intent(in)::comm
end
subroutine write_checkpoint(self,solver,writer_session)
class(stats_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
type(writer_session_t),intent(inout)::writer_session
end
subroutine read_checkpoint(self,solver,reader_session)
class(stats_manager_t),intent(inout)::self
class(solver_t),intent(in)::solver
type(reader_session_t),intent(inout)::reader_session
end
subroutine finalise(self)
class(stats_manager_t),intent(inout)::self
end
end
