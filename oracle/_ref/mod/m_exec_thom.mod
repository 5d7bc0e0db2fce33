﻿!mod$ v1 sum:f42af58fa694e3e8
!need$ 0e4dd7951302c046 n m_tdsops
!need$ 8c38ea408f78ccc8 n m_omp_kernels_thom
!need$ f74ae58d325d162e n m_common
module m_exec_thom
use m_common,only:dp
use m_tdsops,only:tdsops_t
use m_omp_kernels_thom,only:der_univ_thom
use m_omp_kernels_thom,only:der_univ_thom_per
private::dp
private::tdsops_t
private::der_univ_thom
private::der_univ_thom_per
contains
subroutine exec_thom_tds_compact(du,u,tdsops,n_groups)
real(8),intent(out)::du(:,:,:)
real(8),intent(in)::u(:,:,:)
type(tdsops_t),intent(in)::tdsops
integer(4),intent(in)::n_groups
end
end
