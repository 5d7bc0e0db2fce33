﻿!mod$ v1 sum:0e4dd7951302c046
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ f74ae58d325d162e n m_common
module m_tdsops
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_common,only:dp
use m_common,only:pi
use m_common,only:vert
use m_common,only:cell
use m_common,only:bc_periodic
use m_common,only:bc_neumann
use m_common,only:bc_dirichlet
type::tdsops_t
real(8),allocatable::dist_fw(:)
real(8),allocatable::dist_bw(:)
real(8),allocatable::dist_sa(:)
real(8),allocatable::dist_sc(:)
real(8),allocatable::dist_af(:)
real(8),allocatable::thom_f(:)
real(8),allocatable::thom_s(:)
real(8),allocatable::thom_w(:)
real(8),allocatable::thom_p(:)
real(8),allocatable::stretch(:)
real(8),allocatable::stretch_correct(:)
real(8),allocatable::coeffs(:)
real(8),allocatable::coeffs_s(:,:)
real(8),allocatable::coeffs_e(:,:)
real(8)::alpha
real(8)::a
real(8)::b
real(8)::c=0._8
real(8)::d=0._8
real(8)::beta=0._8
real(8)::beta_lhs_s=0._8
logical(4)::periodic
logical(4)::pentadiag=.false._4
integer(4)::n_tds
integer(4)::n_rhs
integer(4)::move=0_4
integer(4)::n_halo
contains
procedure::deriv_1st
procedure::deriv_2nd
procedure::interpl_mid
procedure::stagder_1st
procedure::preprocess_dist
procedure::preprocess_penta_dist
procedure::preprocess_thom
end type
type::dirps_t
class(tdsops_t),allocatable::der1st
class(tdsops_t),allocatable::der1st_sym
class(tdsops_t),allocatable::der2nd
class(tdsops_t),allocatable::der2nd_sym
class(tdsops_t),allocatable::stagder_v2p
class(tdsops_t),allocatable::stagder_p2v
class(tdsops_t),allocatable::interpl_v2p
class(tdsops_t),allocatable::interpl_p2v
integer(4)::dir
end type
interface tdsops_t
procedure::tdsops_init
end interface
contains
function tdsops_init(n_tds,delta,operation,scheme,bc_start,bc_end,stretch,stretch_correct,n_halo,from_to,sym,c_nu,nu0_nu) result(tdsops)
integer(4),intent(in)::n_tds
real(8),intent(in)::delta
character(*,1),intent(in)::operation
character(*,1),intent(in)::scheme
integer(4),intent(in)::bc_start
integer(4),intent(in)::bc_end
real(8),intent(in),optional::stretch(:)
real(8),intent(in),optional::stretch_correct(:)
integer(4),intent(in),optional::n_halo
character(*,1),intent(in),optional::from_to
logical(4),intent(in),optional::sym
real(8),intent(in),optional::c_nu
real(8),intent(in),optional::nu0_nu
type(tdsops_t)::tdsops
end
subroutine deriv_1st(self,delta,scheme,bc_start,bc_end,sym)
class(tdsops_t),intent(inout)::self
real(8),intent(in)::delta
character(*,1),intent(in)::scheme
integer(4),intent(in)::bc_start
integer(4),intent(in)::bc_end
logical(4),intent(in),optional::sym
end
subroutine deriv_2nd(self,delta,scheme,bc_start,bc_end,sym,c_nu,nu0_nu)
class(tdsops_t),intent(inout)::self
real(8),intent(in)::delta
character(*,1),intent(in)::scheme
integer(4),intent(in)::bc_start
integer(4),intent(in)::bc_end
logical(4),intent(in),optional::sym
real(8),intent(in),optional::c_nu
real(8),intent(in),optional::nu0_nu
end
subroutine interpl_mid(self,scheme,from_to,bc_start,bc_end,sym)
class(tdsops_t),intent(inout)::self
character(*,1),intent(in)::scheme
character(*,1),intent(in)::from_to
integer(4),intent(in)::bc_start
integer(4),intent(in)::bc_end
logical(4),intent(in),optional::sym
end
subroutine stagder_1st(self,delta,scheme,from_to,bc_start,bc_end,sym)
class(tdsops_t),intent(inout)::self
real(8),intent(in)::delta
character(*,1),intent(in)::scheme
character(*,1),intent(in)::from_to
integer(4),intent(in)::bc_start
integer(4),intent(in)::bc_end
logical(4),intent(in),optional::sym
end
subroutine preprocess_dist(self,dist_b)
class(tdsops_t),intent(inout)::self
real(8),intent(in)::dist_b(:)
end
subroutine preprocess_thom(self,b)
class(tdsops_t),intent(inout)::self
real(8),intent(in)::b(:)
end
subroutine preprocess_penta_dist(self,bc_start,bc_end,symmetry)
class(tdsops_t),intent(inout)::self
integer(4),intent(in)::bc_start
integer(4),intent(in)::bc_end
logical(4),intent(in)::symmetry
end
end
