﻿!mod$ v1 sum:4aadfe086b475d81
!need$ f39a1ef65bd4689d n m_base_backend
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
!need$ 0e4dd7951302c046 n m_tdsops
!need$ 939e7b51cda90705 n m_allocator
!need$ f1de5abe9bfe2168 i iso_fortran_env
module m_vector_calculus
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_allocator,only:allocator_t
use m_base_backend,only:base_backend_t
use m_common,only:dp
use m_common,only:dir_x
use m_common,only:dir_y
use m_common,only:dir_z
use m_common,only:rdr_x2y
use m_common,only:rdr_x2z
use m_common,only:rdr_y2x
use m_common,only:rdr_y2z
use m_common,only:rdr_z2x
use m_common,only:rdr_z2y
use m_field,only:field_t
use m_tdsops,only:tdsops_t
type::vector_calculus_t
class(base_backend_t),pointer::backend
contains
procedure::curl
procedure::divergence_v2c
procedure::gradient_c2v
procedure::interpl_c2v
procedure::laplacian
end type
interface vector_calculus_t
procedure::init
end interface
contains
function init(backend) result(vector_calculus)
class(base_backend_t),intent(inout),target::backend
type(vector_calculus_t)::vector_calculus
end
subroutine curl(self,o_i_hat,o_j_hat,o_k_hat,u,v,w,x_der1st,y_der1st,z_der1st)
class(vector_calculus_t)::self
class(field_t),intent(inout)::o_i_hat
class(field_t),intent(inout)::o_j_hat
class(field_t),intent(inout)::o_k_hat
class(field_t),intent(in)::u
class(field_t),intent(in)::v
class(field_t),intent(in)::w
class(tdsops_t),intent(in)::x_der1st
class(tdsops_t),intent(in)::y_der1st
class(tdsops_t),intent(in)::z_der1st
end
subroutine divergence_v2c(self,div_u,u,v,w,x_stagder_v2c,x_interpl_v2c,y_stagder_v2c,y_interpl_v2c,z_stagder_v2c,z_interpl_v2c)
class(vector_calculus_t)::self
class(field_t),intent(inout)::div_u
class(field_t),intent(in)::u
class(field_t),intent(in)::v
class(field_t),intent(in)::w
class(tdsops_t),intent(in)::x_stagder_v2c
class(tdsops_t),intent(in)::x_interpl_v2c
class(tdsops_t),intent(in)::y_stagder_v2c
class(tdsops_t),intent(in)::y_interpl_v2c
class(tdsops_t),intent(in)::z_stagder_v2c
class(tdsops_t),intent(in)::z_interpl_v2c
end
subroutine gradient_c2v(self,dpdx,dpdy,dpdz,p,x_stagder_c2v,x_interpl_c2v,y_stagder_c2v,y_interpl_c2v,z_stagder_c2v,z_interpl_c2v)
class(vector_calculus_t)::self
class(field_t),intent(inout)::dpdx
class(field_t),intent(inout)::dpdy
class(field_t),intent(inout)::dpdz
class(field_t),intent(in)::p
class(tdsops_t),intent(in)::x_stagder_c2v
class(tdsops_t),intent(in)::x_interpl_c2v
class(tdsops_t),intent(in)::y_stagder_c2v
class(tdsops_t),intent(in)::y_interpl_c2v
class(tdsops_t),intent(in)::z_stagder_c2v
class(tdsops_t),intent(in)::z_interpl_c2v
end
subroutine interpl_c2v(self,p_out,p,x_interpl_c2v,y_interpl_c2v,z_interpl_c2v)
class(vector_calculus_t)::self
class(field_t),intent(inout)::p_out
class(field_t),intent(in)::p
class(tdsops_t),intent(in)::x_interpl_c2v
class(tdsops_t),intent(in)::y_interpl_c2v
class(tdsops_t),intent(in)::z_interpl_c2v
end
subroutine laplacian(self,lapl_u,u,x_der2nd,y_der2nd,z_der2nd)
class(vector_calculus_t)::self
class(field_t),intent(inout)::lapl_u
class(field_t),intent(in)::u
class(tdsops_t),intent(in)::x_der2nd
class(tdsops_t),intent(in)::y_der2nd
class(tdsops_t),intent(in)::z_der2nd
end
end
