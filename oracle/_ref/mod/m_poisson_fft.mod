﻿!mod$ v1 sum:8bbd3c8ff5c3ea46
!need$ f4f3b1cdb42159bf n m_mesh
!need$ 0e4dd7951302c046 n m_tdsops
!need$ d9a8bda24462498c n m_field
!need$ f74ae58d325d162e n m_common
module m_poisson_fft
use m_common,only:dp
use m_common,only:pi
use m_common,only:cell
use m_field,only:field_t
use m_mesh,only:mesh_t
use m_mesh,only:geo_t
use m_tdsops,only:dirps_t
type,abstract::poisson_fft_t
type(mesh_t),pointer::mesh=>NULL()
integer(4)::nx_glob
integer(4)::ny_glob
integer(4)::nz_glob
integer(4)::nx_loc
integer(4)::ny_loc
integer(4)::nz_loc
integer(4)::nx_perm
integer(4)::ny_perm
integer(4)::nz_perm
integer(4)::nx_spec
integer(4)::ny_spec
integer(4)::nz_spec
integer(4)::sp_st(1_8:3_8)
complex(8),allocatable::waves(:,:,:)
real(8),allocatable::ax(:)
real(8),allocatable::bx(:)
real(8),allocatable::ay(:)
real(8),allocatable::by(:)
real(8),allocatable::az(:)
real(8),allocatable::bz(:)
complex(8),allocatable::kx(:)
complex(8),allocatable::ky(:)
complex(8),allocatable::kz(:)
complex(8),allocatable::exs(:)
complex(8),allocatable::eys(:)
complex(8),allocatable::ezs(:)
complex(8),allocatable::k2x(:)
complex(8),allocatable::k2y(:)
complex(8),allocatable::k2z(:)
real(8),allocatable::trans_x_re(:)
real(8),allocatable::trans_x_im(:)
real(8),allocatable::trans_y_re(:)
real(8),allocatable::trans_y_im(:)
real(8),allocatable::trans_z_re(:)
real(8),allocatable::trans_z_im(:)
logical(4)::periodic_x
logical(4)::periodic_y
logical(4)::periodic_z
logical(4)::stretched_y=.false._4
logical(4)::stretched_y_sym
real(8),allocatable::a_odd_re(:,:,:,:)
real(8),allocatable::a_odd_im(:,:,:,:)
real(8),allocatable::a_even_re(:,:,:,:)
real(8),allocatable::a_even_im(:,:,:,:)
real(8),allocatable::a_re(:,:,:,:)
real(8),allocatable::a_im(:,:,:,:)
logical(4)::lowmem=.false._4
procedure(poisson_xxx),pointer::poisson
contains
procedure(fft_forward),deferred::fft_forward_010
procedure(fft_forward),deferred::fft_forward_100
procedure(fft_forward),deferred::fft_forward_110
procedure(fft_forward),deferred::fft_forward
procedure(fft_backward),deferred::fft_backward_010
procedure(fft_backward),deferred::fft_backward_100
procedure(fft_backward),deferred::fft_backward_110
procedure(fft_backward),deferred::fft_backward
procedure(fft_postprocess),deferred::fft_postprocess_000
procedure(fft_postprocess),deferred::fft_postprocess_010
procedure(fft_postprocess),deferred::fft_postprocess_100
procedure(fft_postprocess),deferred::fft_postprocess_110
procedure(field_process),deferred::enforce_periodicity_x
procedure(field_process),deferred::undo_periodicity_x
procedure(field_process),deferred::enforce_periodicity_y
procedure(field_process),deferred::undo_periodicity_y
procedure(field_process),deferred::enforce_periodicity_xy
procedure(field_process),deferred::undo_periodicity_xy
procedure::base_init
procedure::solve_poisson
procedure::stretching_matrix
procedure::waves_set
procedure::get_km
procedure::get_km_re
procedure::get_km_im
end type
intrinsic::null
abstract interface
subroutine fft_forward(self,f_in)
import::field_t
import::poisson_fft_t
class(poisson_fft_t)::self
class(field_t),intent(in)::f_in
end
end interface
abstract interface
subroutine fft_backward(self,f_out)
import::field_t
import::poisson_fft_t
class(poisson_fft_t)::self
class(field_t),intent(inout)::f_out
end
end interface
abstract interface
subroutine fft_postprocess(self)
import::poisson_fft_t
class(poisson_fft_t)::self
end
end interface
abstract interface
subroutine poisson_xxx(self,f,temp)
import::field_t
import::poisson_fft_t
class(poisson_fft_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(inout)::temp
end
end interface
abstract interface
subroutine field_process(self,f_out,f_in)
import::field_t
import::poisson_fft_t
class(poisson_fft_t)::self
class(field_t),intent(inout)::f_out
class(field_t),intent(in)::f_in
end
end interface
contains
subroutine base_init(self,mesh,xdirps,ydirps,zdirps,n_spec,n_sp_st)
class(poisson_fft_t)::self
type(mesh_t),intent(in),target::mesh
type(dirps_t),intent(in)::xdirps
type(dirps_t),intent(in)::ydirps
type(dirps_t),intent(in)::zdirps
integer(4),intent(in)::n_spec(1_8:3_8)
integer(4),intent(in)::n_sp_st(1_8:3_8)
end
subroutine solve_poisson(self,f,temp)
class(poisson_fft_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(inout)::temp
end
subroutine poisson_000(self,f,temp)
class(poisson_fft_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(inout)::temp
end
subroutine poisson_010(self,f,temp)
class(poisson_fft_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(inout)::temp
end
subroutine poisson_100(self,f,temp)
class(poisson_fft_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(inout)::temp
end
subroutine poisson_110(self,f,temp)
class(poisson_fft_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(inout)::temp
end
subroutine stretching_matrix(self,geo,xdirps,ydirps,zdirps)
class(poisson_fft_t)::self
type(geo_t),intent(in)::geo
type(dirps_t),intent(in)::xdirps
type(dirps_t),intent(in)::ydirps
type(dirps_t),intent(in)::zdirps
end
subroutine waves_set(self,geo,xdirps,ydirps,zdirps)
class(poisson_fft_t)::self
type(geo_t),intent(in)::geo
type(dirps_t),intent(in)::xdirps
type(dirps_t),intent(in)::ydirps
type(dirps_t),intent(in)::zdirps
end
subroutine wave_numbers(a,b,k,e,k2,n,l,d,periodic,c_a,c_b,c_alpha)
real(8),intent(out)::a(:)
real(8),intent(out)::b(:)
complex(8),intent(out)::k(:)
complex(8),intent(out)::e(:)
complex(8),intent(out)::k2(:)
integer(4),intent(in)::n
real(8),intent(in)::l
real(8),intent(in)::d
logical(4),intent(in)::periodic
real(8),intent(in)::c_a
real(8),intent(in)::c_b
real(8),intent(in)::c_alpha
end
function get_km_re(self,i,j,k) result(re)
class(poisson_fft_t)::self
integer(4),intent(in)::i
integer(4),intent(in)::j
integer(4),intent(in)::k
real(8)::re
end
function get_km_im(self,i,j,k) result(re)
class(poisson_fft_t)::self
integer(4),intent(in)::i
integer(4),intent(in)::j
integer(4),intent(in)::k
real(8)::re
end
function get_km(self,i,j,k) result(km)
class(poisson_fft_t)::self
integer(4),intent(in)::i
integer(4),intent(in)::j
integer(4),intent(in)::k
complex(8)::km
end
function get_real(complx) result(re)
complex(8),intent(in)::complx
real(8)::re
end
function get_imag(complx) result(im)
complex(8),intent(in)::complx
real(8)::im
end
end
