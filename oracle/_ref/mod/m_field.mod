﻿!mod$ v1 sum:d9a8bda24462498c
!need$ f74ae58d325d162e n m_common
module m_field
use m_common,only:dp
use m_common,only:dir_x
use m_common,only:dir_y
use m_common,only:dir_z
use m_common,only:dir_c
type::field_t
class(field_t),pointer::next
real(8),pointer,private::p_data(:)
real(8),contiguous,pointer::data(:,:,:)
integer(4)::dir
integer(4)::data_loc
integer(4)::refcount=0_4
integer(4)::id
contains
procedure::fill
procedure::get_shape
procedure::set_shape
procedure::set_data_loc
end type
type::flist_t
class(field_t),pointer::ptr
end type
interface field_t
procedure::field_init
end interface
contains
function field_init(ngrid,next,id) result(f)
integer(4),intent(in)::ngrid
type(field_t),intent(in),pointer::next
integer(4),intent(in)::id
type(field_t)::f
end
subroutine fill(self,c)
class(field_t)::self
real(8),intent(in)::c
end
subroutine set_data_loc(self,data_loc)
class(field_t)::self
integer(4),intent(in)::data_loc
end
function get_shape(self) result(dims)
class(field_t)::self
integer(4)::dims(1_8:3_8)
end
subroutine set_shape(self,dims)
class(field_t)::self
integer(4),intent(in)::dims(1_8:3_8)
end
end
