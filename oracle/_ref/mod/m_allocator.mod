﻿!mod$ v1 sum:939e7b51cda90705
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ d9a8bda24462498c n m_field
!need$ f74ae58d325d162e n m_common
module m_allocator
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_common,only:dp
use m_common,only:dir_x
use m_common,only:dir_y
use m_common,only:dir_z
use m_common,only:dir_c
use m_common,only:null_loc
use m_field,only:field_t
use m_field,only:m_field$m_field$field_init=>field_init
type::allocator_t
integer(4)::ngrid
integer(4)::sz
integer(4)::next_id=0_4
integer(4),private::dims_padded_dir(1_8:3_8,1_8:4_8)
integer(4),private::n_groups_dir(1_8:3_8)
class(field_t),pointer::first=>NULL()
contains
procedure::get_block
procedure::release_block
procedure::create_block
procedure::get_block_ids
procedure::destroy
procedure::get_padded_dims
procedure::get_n_groups
end type
intrinsic::null
interface allocator_t
procedure::allocator_init
end interface
contains
function allocator_init(dims,sz) result(allocator)
integer(4),intent(in)::dims(1_8:3_8)
integer(4),intent(in)::sz
type(allocator_t)::allocator
end
function create_block(self,next) result(ptr)
class(allocator_t),intent(inout)::self
class(field_t),intent(in),pointer::next
class(field_t),pointer::ptr
end
function get_block(self,dir,data_loc) result(handle)
class(allocator_t),intent(inout)::self
integer(4),intent(in)::dir
integer(4),intent(in),optional::data_loc
class(field_t),pointer::handle
end
subroutine release_block(self,handle)
class(allocator_t),intent(inout)::self
class(field_t),pointer::handle
end
subroutine destroy(self)
class(allocator_t),intent(inout)::self
end
function get_block_ids(self)
class(allocator_t),intent(inout)::self
integer(4),allocatable::get_block_ids(:)
end
function get_padded_dims(self,dir) result(dims)
class(allocator_t),intent(inout)::self
integer(4),intent(in)::dir
integer(4)::dims(1_8:3_8)
end
function get_n_groups(self,dir) result(n_groups)
class(allocator_t),intent(inout)::self
integer(4),intent(in)::dir
integer(4)::n_groups
end
end
