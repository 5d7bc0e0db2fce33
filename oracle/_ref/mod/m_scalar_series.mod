﻿!mod$ v1 sum:4e78cfb6ee5c840a
!need$ f74ae58d325d162e n m_common
module m_scalar_series
use m_common,only:dp
private::dp
type::scalar_series_t
integer(4),private::file_unit=-1_4
integer(4),private::n_columns=0_4
logical(4),private::is_root=.false._4
contains
procedure::init
procedure::write_step
procedure::finalise
end type
private::init
private::write_step
private::finalise
contains
subroutine init(self,filename,column_names,is_root,append)
class(scalar_series_t),intent(inout)::self
character(*,1),intent(in)::filename
character(*,1),intent(in)::column_names(:)
logical(4),intent(in)::is_root
logical(4),intent(in)::append
end
subroutine write_step(self,t,values)
class(scalar_series_t),intent(inout)::self
real(8),intent(in)::t
real(8),intent(in)::values(:)
end
subroutine finalise(self)
class(scalar_series_t),intent(inout)::self
end
end
