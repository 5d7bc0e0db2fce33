﻿!mod$ v1 sum:740fa3474af256b4
!need$ f74ae58d325d162e n m_common
module m_ordering
use m_common,only:dp
use m_common,only:get_dirs_from_rdr
use m_common,only:dir_x
use m_common,only:dir_y
use m_common,only:dir_z
use m_common,only:dir_c
contains
pure subroutine get_index_ijk(i,j,k,dir_i,dir_j,dir_k,dir,sz,nx_padded,ny_padded,nz_padded)
integer(4),intent(out)::i
integer(4),intent(out)::j
integer(4),intent(out)::k
integer(4),intent(in)::dir_i
integer(4),intent(in)::dir_j
integer(4),intent(in)::dir_k
integer(4),intent(in)::dir
integer(4),intent(in)::sz
integer(4),intent(in)::nx_padded
integer(4),intent(in)::ny_padded
integer(4),intent(in)::nz_padded
end
pure subroutine get_index_dir(dir_i,dir_j,dir_k,i,j,k,dir,sz,nx_padded,ny_padded,nz_padded)
integer(4),intent(out)::dir_i
integer(4),intent(out)::dir_j
integer(4),intent(out)::dir_k
integer(4),intent(in)::i
integer(4),intent(in)::j
integer(4),intent(in)::k
integer(4),intent(in)::dir
integer(4),intent(in)::sz
integer(4),intent(in)::nx_padded
integer(4),intent(in)::ny_padded
integer(4),intent(in)::nz_padded
end
pure subroutine get_index_reordering(out_i,out_j,out_k,in_i,in_j,in_k,dir_from,dir_to,sz,cart_padded)
integer(4),intent(out)::out_i
integer(4),intent(out)::out_j
integer(4),intent(out)::out_k
integer(4),intent(in)::in_i
integer(4),intent(in)::in_j
integer(4),intent(in)::in_k
integer(4),intent(in)::dir_from
integer(4),intent(in)::dir_to
integer(4),intent(in)::sz
integer(4),intent(in)::cart_padded(1_8:3_8)
end
end
