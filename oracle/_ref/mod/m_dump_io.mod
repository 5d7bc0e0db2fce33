﻿!mod$ v1 sum:480480b6304b6ecc
!need$ f74ae58d325d162e n m_common
module m_dump_io
use m_common,only:dp
integer(4)::dump_unit
contains
subroutine dump_open(fname)
character(*,1),intent(in)::fname
end
subroutine dump_close()
end
subroutine dump_hdr(name,rank,dims)
character(*,1),intent(in)::name
integer(4),intent(in)::rank
integer(4),intent(in)::dims(:)
end
subroutine dump_r1(name,a)
character(*,1),intent(in)::name
real(8),intent(in)::a(:)
end
subroutine dump_r2(name,a)
character(*,1),intent(in)::name
real(8),intent(in)::a(:,:)
end
subroutine dump_r3(name,a)
character(*,1),intent(in)::name
real(8),intent(in)::a(:,:,:)
end
subroutine dump_i1(name,a)
character(*,1),intent(in)::name
integer(4),intent(in)::a(:)
end
subroutine dump_s(name,x)
character(*,1),intent(in)::name
real(8),intent(in)::x
end
end
