﻿!mod$ v1 sum:8c38ea408f78ccc8
!need$ f74ae58d325d162e n m_common
!need$ 5aa2ec0b70f94be2 n m_omp_common
module m_omp_kernels_thom
use m_common,only:dp
use m_omp_common,only:sz
contains
subroutine der_univ_thom(du,u,n_tds,n_rhs,coeffs_s,coeffs_e,coeffs,thom_f,thom_s,thom_w,strch)
real(8),intent(out)::du(:,:)
real(8),intent(in)::u(:,:)
integer(4),intent(in)::n_tds
integer(4),intent(in)::n_rhs
real(8),intent(in)::coeffs_s(:,:)
real(8),intent(in)::coeffs_e(:,:)
real(8),intent(in)::coeffs(:)
real(8),intent(in)::thom_f(:)
real(8),intent(in)::thom_s(:)
real(8),intent(in)::thom_w(:)
real(8),intent(in)::strch(:)
end
subroutine der_univ_thom_per(du,u,n,coeffs,alpha,thom_f,thom_s,thom_w,thom_p,strch)
real(8),intent(out)::du(:,:)
real(8),intent(in)::u(:,:)
integer(4),intent(in)::n
real(8),intent(in)::coeffs(:)
real(8),intent(in)::alpha
real(8),intent(in)::thom_f(:)
real(8),intent(in)::thom_s(:)
real(8),intent(in)::thom_w(:)
real(8),intent(in)::thom_p(:)
real(8),intent(in)::strch(:)
end
end
