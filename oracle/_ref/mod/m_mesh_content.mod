﻿!mod$ v1 sum:ef9e877cc44b415d
!need$ f74ae58d325d162e n m_common
module m_mesh_content
use m_common,only:dp
use m_common,only:pi
type::geo_t
real(8)::origin(1_8:3_8)
real(8)::d(1_8:3_8)
real(8)::l(1_8:3_8)
real(8),allocatable::vert_coords(:,:)
real(8),allocatable::midp_coords(:,:)
character(20_4,1)::stretching(1_8:3_8)
logical(4)::stretched(1_8:3_8)
real(8)::alpha(1_8:3_8)
real(8)::beta(1_8:3_8)
real(8),allocatable::vert_ds(:,:)
real(8),allocatable::vert_ds2(:,:)
real(8),allocatable::vert_d2s(:,:)
real(8),allocatable::midp_ds(:,:)
real(8),allocatable::midp_ds2(:,:)
real(8),allocatable::midp_d2s(:,:)
contains
procedure::obtain_coordinates
end type
type::grid_t
integer(4)::global_vert_dims(1_8:3_8)
integer(4)::global_cell_dims(1_8:3_8)
integer(4)::vert_dims(1_8:3_8)
integer(4)::cell_dims(1_8:3_8)
logical(4)::periodic_bc(1_8:3_8)
integer(4)::bcs_global(1_8:3_8,1_8:2_8)
integer(4)::bcs(1_8:3_8,1_8:2_8)
contains
procedure::copy_cell2vert_dims
procedure::copy_vert2cell_dims
end type
type::par_t
integer(4)::nrank
integer(4)::nproc
integer(4)::nrank_dir(1_8:3_8)
integer(4)::nproc_dir(1_8:3_8)
integer(4)::n_offset(1_8:3_8)
integer(4)::pnext(1_8:3_8)
integer(4)::pprev(1_8:3_8)
contains
procedure::is_root
procedure::compute_rank_pos_from_global
end type
contains
pure function is_root(self) result(is_root_rank)
class(par_t),intent(in)::self
logical(4)::is_root_rank
end
pure subroutine compute_rank_pos_from_global(self,global_ranks)
class(par_t),intent(inout)::self
integer(4),intent(in)::global_ranks(:,:,:)
end
pure subroutine copy_vert2cell_dims(self,par)
class(grid_t),intent(inout)::self
type(par_t),intent(in)::par
end
pure subroutine copy_cell2vert_dims(self,par)
class(grid_t),intent(inout)::self
type(par_t),intent(in)::par
end
subroutine obtain_coordinates(self,vert_dims,cell_dims,n_offset)
class(geo_t)::self
integer(4),intent(in)::vert_dims(1_8:3_8)
integer(4),intent(in)::cell_dims(1_8:3_8)
integer(4),intent(in)::n_offset(1_8:3_8)
end
end
