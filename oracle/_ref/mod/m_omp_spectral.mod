﻿!mod$ v1 sum:0b3b2a777c44597d
!need$ f74ae58d325d162e n m_common
module m_omp_spectral
use m_common,only:dp
contains
subroutine process_spectral_000(div_u,waves,nx_spec,ny_spec,nz_spec,x_sp_st,y_sp_st,z_sp_st,nx,ny,nz,ax,bx,ay,by,az,bz)
complex(8),intent(inout)::div_u(:,:,:)
complex(8),intent(in)::waves(:,:,:)
integer(4),intent(in)::nx_spec
integer(4),intent(in)::ny_spec
integer(4),intent(in)::nz_spec
integer(4),intent(in)::x_sp_st
integer(4),intent(in)::y_sp_st
integer(4),intent(in)::z_sp_st
integer(4),intent(in)::nx
integer(4),intent(in)::ny
integer(4),intent(in)::nz
real(8),intent(in)::ax(:)
real(8),intent(in)::bx(:)
real(8),intent(in)::ay(:)
real(8),intent(in)::by(:)
real(8),intent(in)::az(:)
real(8),intent(in)::bz(:)
end
subroutine process_spectral_010(div_u,waves,nx_spec,ny_spec,nz_spec,x_sp_st,y_sp_st,z_sp_st,nx,ny,nz,ax,bx,ay,by,az,bz)
complex(8),intent(inout)::div_u(:,:,:)
complex(8),intent(in)::waves(:,:,:)
integer(4),intent(in)::nx_spec
integer(4),intent(in)::ny_spec
integer(4),intent(in)::nz_spec
integer(4),intent(in)::x_sp_st
integer(4),intent(in)::y_sp_st
integer(4),intent(in)::z_sp_st
integer(4),intent(in)::nx
integer(4),intent(in)::ny
integer(4),intent(in)::nz
real(8),intent(in)::ax(:)
real(8),intent(in)::bx(:)
real(8),intent(in)::ay(:)
real(8),intent(in)::by(:)
real(8),intent(in)::az(:)
real(8),intent(in)::bz(:)
end
end
