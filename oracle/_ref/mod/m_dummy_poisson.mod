﻿!mod$ v1 sum:0ad26db9e8a04759
!need$ 8bbd3c8ff5c3ea46 n m_poisson_fft
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
module m_dummy_poisson
use m_common,only:dp
use m_field,only:field_t
use m_poisson_fft,only:poisson_fft_t
type,extends(poisson_fft_t)::dummy_poisson_t
contains
procedure::fft_forward_010=>fw
procedure::fft_forward_100=>fw
procedure::fft_forward_110=>fw
procedure::fft_forward=>fw
procedure::fft_backward_010=>bw
procedure::fft_backward_100=>bw
procedure::fft_backward_110=>bw
procedure::fft_backward=>bw
procedure::fft_postprocess_000=>pp
procedure::fft_postprocess_010=>pp
procedure::fft_postprocess_100=>pp
procedure::fft_postprocess_110=>pp
procedure::enforce_periodicity_x=>fp
procedure::undo_periodicity_x=>fp
procedure::enforce_periodicity_y=>fp
procedure::undo_periodicity_y=>fp
procedure::enforce_periodicity_xy=>fp
procedure::undo_periodicity_xy=>fp
end type
contains
subroutine fw(self,f_in)
class(dummy_poisson_t)::self
class(field_t),intent(in)::f_in
end
subroutine bw(self,f_out)
class(dummy_poisson_t)::self
class(field_t),intent(inout)::f_out
end
subroutine pp(self)
class(dummy_poisson_t)::self
end
subroutine fp(self,f_out,f_in)
class(dummy_poisson_t)::self
class(field_t),intent(inout)::f_out
class(field_t),intent(in)::f_in
end
end
