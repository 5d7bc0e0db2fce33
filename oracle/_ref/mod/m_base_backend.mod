﻿!mod$ v1 sum:f39a1ef65bd4689d
!need$ 939e7b51cda90705 n m_allocator
!need$ 8bbd3c8ff5c3ea46 n m_poisson_fft
!need$ 0e4dd7951302c046 n m_tdsops
!need$ 0df96a70750958ab n mpi
!need$ d9a8bda24462498c n m_field
!need$ f4f3b1cdb42159bf n m_mesh
!need$ f74ae58d325d162e n m_common
module m_base_backend
use m_allocator,only:allocator_t
use m_common,only:dp
use m_common,only:dir_c
use m_common,only:get_rdr_from_dirs
use m_field,only:field_t
use m_mesh,only:mesh_t
use m_poisson_fft,only:poisson_fft_t
use m_tdsops,only:tdsops_t
use m_tdsops,only:dirps_t
use mpi,only:mpi_source
use mpi,only:mpi_tag
use mpi,only:mpi_error
use mpi,only:mpi_status_size
use mpi,only:mpi_success
use mpi,only:mpi_err_other
use mpi,only:mpi_err_count
use mpi,only:mpi_err_spawn
use mpi,only:mpi_err_locktype
use mpi,only:mpi_err_op
use mpi,only:mpi_err_dup_datarep
use mpi,only:mpi_err_unsupported_datarep
use mpi,only:mpi_err_truncate
use mpi,only:mpi_err_info_nokey
use mpi,only:mpi_err_assert
use mpi,only:mpi_err_file_exists
use mpi,only:mpi_err_pending
use mpi,only:mpi_err_comm
use mpi,only:mpi_err_keyval
use mpi,only:mpi_err_name
use mpi,only:mpi_err_request
use mpi,only:mpi_err_type
use mpi,only:mpi_err_info_value
use mpi,only:mpi_err_rma_sync
use mpi,only:mpi_err_no_mem
use mpi,only:mpi_err_bad_file
use mpi,only:mpi_err_quota
use mpi,only:mpi_err_root
use mpi,only:mpi_err_service
use mpi,only:mpi_err_io
use mpi,only:mpi_err_rma_flavor
use mpi,only:mpi_err_access
use mpi,only:mpi_err_no_space
use mpi,only:mpi_err_conversion
use mpi,only:mpi_err_win
use mpi,only:mpi_err_file
use mpi,only:mpi_err_rma_shared
use mpi,only:mpi_err_base
use mpi,only:mpi_err_rma_conflict
use mpi,only:mpi_err_in_status
use mpi,only:mpi_err_info_key
use mpi,only:mpi_err_arg
use mpi,only:mpi_err_read_only
use mpi,only:mpi_err_size
use mpi,only:mpi_err_buffer
use mpi,only:mpi_err_lastcode
use mpi,only:mpi_err_disp
use mpi,only:mpi_err_port
use mpi,only:mpi_err_group
use mpi,only:mpi_err_topology
use mpi,only:mpi_err_tag
use mpi,only:mpi_err_not_same
use mpi,only:mpi_err_info
use mpi,only:mpi_err_unknown
use mpi,only:mpi_err_file_in_use
use mpi,only:mpi_err_rma_attach
use mpi,only:mpi_err_unsupported_operation
use mpi,only:mpi_err_amode
use mpi,only:mpi_err_rank
use mpi,only:mpi_err_dims
use mpi,only:mpi_err_no_such_file
use mpi,only:mpi_err_rma_range
use mpi,only:mpi_err_intern
use mpi,only:mpi_errors_are_fatal
use mpi,only:mpi_errors_return
use mpi,only:mpi_ident
use mpi,only:mpi_congruent
use mpi,only:mpi_similar
use mpi,only:mpi_unequal
use mpi,only:mpi_win_flavor_create
use mpi,only:mpi_win_flavor_allocate
use mpi,only:mpi_win_flavor_dynamic
use mpi,only:mpi_win_flavor_shared
use mpi,only:mpi_win_separate
use mpi,only:mpi_win_unified
use mpi,only:mpi_max
use mpi,only:mpi_min
use mpi,only:mpi_sum
use mpi,only:mpi_prod
use mpi,only:mpi_land
use mpi,only:mpi_band
use mpi,only:mpi_lor
use mpi,only:mpi_bor
use mpi,only:mpi_lxor
use mpi,only:mpi_bxor
use mpi,only:mpi_minloc
use mpi,only:mpi_maxloc
use mpi,only:mpi_replace
use mpi,only:mpi_no_op
use mpi,only:mpi_comm_world
use mpi,only:mpi_comm_self
use mpi,only:mpi_group_empty
use mpi,only:mpi_comm_null
use mpi,only:mpi_win_null
use mpi,only:mpi_file_null
use mpi,only:mpi_group_null
use mpi,only:mpi_op_null
use mpi,only:mpi_datatype_null
use mpi,only:mpi_request_null
use mpi,only:mpi_errhandler_null
use mpi,only:mpi_info_null
use mpi,only:mpi_info_env
use mpi,only:mpi_tag_ub
use mpi,only:mpi_host
use mpi,only:mpi_io
use mpi,only:mpi_wtime_is_global
use mpi,only:mpi_universe_size
use mpi,only:mpi_lastusedcode
use mpi,only:mpi_appnum
use mpi,only:mpi_win_base
use mpi,only:mpi_win_size
use mpi,only:mpi_win_disp_unit
use mpi,only:mpi_win_create_flavor
use mpi,only:mpi_win_model
use mpi,only:mpi_max_error_string
use mpi,only:mpi_max_port_name
use mpi,only:mpi_max_object_name
use mpi,only:mpi_max_info_key
use mpi,only:mpi_max_info_val
use mpi,only:mpi_max_processor_name
use mpi,only:mpi_max_datarep_string
use mpi,only:mpi_max_library_version_string
use mpi,only:mpi_undefined
use mpi,only:mpi_keyval_invalid
use mpi,only:mpi_bsend_overhead
use mpi,only:mpi_proc_null
use mpi,only:mpi_any_source
use mpi,only:mpi_any_tag
use mpi,only:mpi_root
use mpi,only:mpi_graph
use mpi,only:mpi_cart
use mpi,only:mpi_dist_graph
use mpi,only:mpi_version
use mpi,only:mpi_subversion
use mpi,only:mpi_lock_exclusive
use mpi,only:mpi_lock_shared
use mpi,only:mpi_complex
use mpi,only:mpi_double_complex
use mpi,only:mpi_logical
use mpi,only:mpi_real
use mpi,only:mpi_double_precision
use mpi,only:mpi_integer
use mpi,only:mpi_2integer
use mpi,only:mpi_2double_precision
use mpi,only:mpi_2real
use mpi,only:mpi_character
use mpi,only:mpi_byte
use mpi,only:mpi_ub
use mpi,only:mpi_lb
use mpi,only:mpi_packed
use mpi,only:mpi_integer1
use mpi,only:mpi_integer2
use mpi,only:mpi_integer4
use mpi,only:mpi_integer8
use mpi,only:mpi_integer16
use mpi,only:mpi_real4
use mpi,only:mpi_real8
use mpi,only:mpi_real16
use mpi,only:mpi_complex8
use mpi,only:mpi_complex16
use mpi,only:mpi_complex32
use mpi,only:mpi_address_kind
use mpi,only:mpi_offset_kind
use mpi,only:mpi_count_kind
use mpi,only:mpi_integer_kind
use mpi,only:mpi_char
use mpi,only:mpi_signed_char
use mpi,only:mpi_unsigned_char
use mpi,only:mpi_wchar
use mpi,only:mpi_short
use mpi,only:mpi_unsigned_short
use mpi,only:mpi_int
use mpi,only:mpi_unsigned
use mpi,only:mpi_long
use mpi,only:mpi_unsigned_long
use mpi,only:mpi_float
use mpi,only:mpi_double
use mpi,only:mpi_long_double
use mpi,only:mpi_long_long_int
use mpi,only:mpi_unsigned_long_long
use mpi,only:mpi_long_long
use mpi,only:mpi_float_int
use mpi,only:mpi_double_int
use mpi,only:mpi_long_int
use mpi,only:mpi_short_int
use mpi,only:mpi_2int
use mpi,only:mpi_long_double_int
use mpi,only:mpi_int8_t
use mpi,only:mpi_int16_t
use mpi,only:mpi_int32_t
use mpi,only:mpi_int64_t
use mpi,only:mpi_uint8_t
use mpi,only:mpi_uint16_t
use mpi,only:mpi_uint32_t
use mpi,only:mpi_uint64_t
use mpi,only:mpi_c_bool
use mpi,only:mpi_c_float_complex
use mpi,only:mpi_c_complex
use mpi,only:mpi_c_double_complex
use mpi,only:mpi_c_long_double_complex
use mpi,only:mpi_aint
use mpi,only:mpi_offset
use mpi,only:mpi_count
use mpi,only:mpi_cxx_bool
use mpi,only:mpi_cxx_float_complex
use mpi,only:mpi_cxx_double_complex
use mpi,only:mpi_cxx_long_double_complex
use mpi,only:mpi_combiner_named
use mpi,only:mpi_combiner_dup
use mpi,only:mpi_combiner_contiguous
use mpi,only:mpi_combiner_vector
use mpi,only:mpi_combiner_hvector_integer
use mpi,only:mpi_combiner_hvector
use mpi,only:mpi_combiner_indexed
use mpi,only:mpi_combiner_hindexed_integer
use mpi,only:mpi_combiner_hindexed
use mpi,only:mpi_combiner_indexed_block
use mpi,only:mpi_combiner_struct_integer
use mpi,only:mpi_combiner_struct
use mpi,only:mpi_combiner_subarray
use mpi,only:mpi_combiner_darray
use mpi,only:mpi_combiner_f90_real
use mpi,only:mpi_combiner_f90_complex
use mpi,only:mpi_combiner_f90_integer
use mpi,only:mpi_combiner_resized
use mpi,only:mpi_combiner_hindexed_block
use mpi,only:mpi_typeclass_real
use mpi,only:mpi_typeclass_integer
use mpi,only:mpi_typeclass_complex
use mpi,only:mpi_mode_nocheck
use mpi,only:mpi_mode_nostore
use mpi,only:mpi_mode_noput
use mpi,only:mpi_mode_noprecede
use mpi,only:mpi_mode_nosucceed
use mpi,only:mpi_comm_type_shared
use mpi,only:mpi_message_null
use mpi,only:mpi_message_no_proc
use mpi,only:mpi_thread_single
use mpi,only:mpi_thread_funneled
use mpi,only:mpi_thread_serialized
use mpi,only:mpi_thread_multiple
use mpi,only:mpi_mode_rdonly
use mpi,only:mpi_mode_rdwr
use mpi,only:mpi_mode_wronly
use mpi,only:mpi_mode_delete_on_close
use mpi,only:mpi_mode_unique_open
use mpi,only:mpi_mode_create
use mpi,only:mpi_mode_excl
use mpi,only:mpi_mode_append
use mpi,only:mpi_mode_sequential
use mpi,only:mpi_seek_set
use mpi,only:mpi_seek_cur
use mpi,only:mpi_seek_end
use mpi,only:mpi_order_c
use mpi,only:mpi_order_fortran
use mpi,only:mpi_distribute_block
use mpi,only:mpi_distribute_cyclic
use mpi,only:mpi_distribute_none
use mpi,only:mpi_distribute_dflt_darg
use mpi,only:mpi_displacement_current
use mpi,only:mpi_subarrays_supported
use mpi,only:mpi_async_protects_nonblocking
use mpi,only:mpi_dup_fn
use mpi,only:mpi_null_delete_fn
use mpi,only:mpi_null_copy_fn
use mpi,only:mpi_comm_dup_fn
use mpi,only:mpi_comm_null_delete_fn
use mpi,only:mpi_comm_null_copy_fn
use mpi,only:mpi_win_dup_fn
use mpi,only:mpi_win_null_delete_fn
use mpi,only:mpi_win_null_copy_fn
use mpi,only:mpi_type_dup_fn
use mpi,only:mpi_type_null_delete_fn
use mpi,only:mpi_type_null_copy_fn
use mpi,only:mpi_conversion_fn_null
use mpi,only:mpi_wtime
use mpi,only:mpi_wtick
use mpi,only:pmpi_wtime
use mpi,only:pmpi_wtick
use mpi,only:mpi_comm_rank
use mpi,only:mpi_comm_size
use mpi,only:mpi_abort
use mpi,only:mpi_reduce
use mpi,only:mpi_initialized
use mpi,only:mpi_unweighted
use mpi,only:mpi_weights_empty
use mpi,only:mpi_bottom
use mpi,only:mpi_in_place
use mpi,only:mpi_status_ignore
use mpi,only:mpi_statuses_ignore
use mpi,only:mpi_errcodes_ignore
use mpi,only:mpi_argvs_null
use mpi,only:mpi_argv_null
type,abstract::base_backend_t
integer(4)::n_halo=4_4
type(mesh_t),pointer::mesh
class(allocator_t),pointer::allocator
class(poisson_fft_t),pointer::poisson_fft
contains
procedure(transeq_ders),deferred::transeq_x
procedure(transeq_ders),deferred::transeq_y
procedure(transeq_ders),deferred::transeq_z
procedure(transeq_ders_spec),deferred::transeq_species
procedure(tds_solve),deferred::tds_solve
procedure(reorder),deferred::reorder
procedure(sum_intox),deferred::sum_yintox
procedure(sum_intox),deferred::sum_zintox
procedure(veccopy),deferred::veccopy
procedure(vecadd),deferred::vecadd
procedure(vecmult),deferred::vecmult
procedure(scalar_product),deferred::scalar_product
procedure(field_max_mean),deferred::field_max_mean
procedure(slice_max_sum),deferred::slice_max_sum
procedure(field_ops),deferred::field_scale
procedure(field_ops),deferred::field_shift
procedure(field_reduce),deferred::field_volume_integral
procedure(field_set_face),deferred::field_set_face
procedure(field_set_face_from_field),deferred::field_set_face_from_field
procedure(derive_field_from_gradients),deferred::compute_vorticity
procedure(derive_field_from_gradients),deferred::compute_qcriterion
procedure(copy_data_to_f),deferred::copy_data_to_f
procedure(copy_f_to_data),deferred::copy_f_to_data
procedure(alloc_tdsops),deferred::alloc_tdsops
procedure(init_poisson_fft),deferred::init_poisson_fft
procedure::base_init
procedure::get_field_data
procedure::set_field_data
end type
abstract interface
subroutine transeq_ders(self,du,dv,dw,u,v,w,nu,dirps)
import::base_backend_t
import::dirps_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::du
class(field_t),intent(inout)::dv
class(field_t),intent(inout)::dw
class(field_t),intent(in)::u
class(field_t),intent(in)::v
class(field_t),intent(in)::w
real(8),intent(in)::nu
type(dirps_t),intent(in)::dirps
end
end interface
abstract interface
subroutine transeq_ders_spec(self,dspec,uvw,spec,nu,dirps,sync)
import::base_backend_t
import::dirps_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::dspec
class(field_t),intent(in)::uvw
class(field_t),intent(in)::spec
real(8),intent(in)::nu
type(dirps_t),intent(in)::dirps
logical(4),intent(in)::sync
end
end interface
abstract interface
subroutine tds_solve(self,du,u,tdsops)
import::base_backend_t
import::field_t
import::tdsops_t
class(base_backend_t)::self
class(field_t),intent(inout)::du
class(field_t),intent(in)::u
class(tdsops_t),intent(in)::tdsops
end
end interface
abstract interface
subroutine reorder(self,u_,u,direction)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::u_
class(field_t),intent(in)::u
integer(4),intent(in)::direction
end
end interface
abstract interface
subroutine sum_intox(self,u,u_)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::u
class(field_t),intent(in)::u_
end
end interface
abstract interface
subroutine veccopy(self,dst,src)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::dst
class(field_t),intent(in)::src
end
end interface
abstract interface
subroutine vecadd(self,a,x,b,y)
import::base_backend_t
import::field_t
class(base_backend_t)::self
real(8),intent(in)::a
class(field_t),intent(in)::x
real(8),intent(in)::b
class(field_t),intent(inout)::y
end
end interface
abstract interface
subroutine vecmult(self,y,x)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::y
class(field_t),intent(in)::x
end
end interface
abstract interface
function scalar_product(self,x,y) result(s)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(in)::x
class(field_t),intent(in)::y
real(8)::s
end
end interface
abstract interface
subroutine field_ops(self,f,a)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(in)::f
real(8),intent(in)::a
end
end interface
abstract interface
function field_reduce(self,f) result(s)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(in)::f
real(8)::s
end
end interface
abstract interface
subroutine field_max_mean(self,max_val,mean_val,f,enforced_data_loc)
import::base_backend_t
import::field_t
class(base_backend_t)::self
real(8),intent(out)::max_val
real(8),intent(out)::mean_val
class(field_t),intent(in)::f
integer(4),intent(in),optional::enforced_data_loc
end
end interface
abstract interface
subroutine slice_max_sum(self,max_val,sum_val,f,i_slice,enforced_data_loc)
import::base_backend_t
import::field_t
class(base_backend_t)::self
real(8),intent(out)::max_val
real(8),intent(out)::sum_val
class(field_t),intent(in)::f
integer(4),intent(in)::i_slice
integer(4),intent(in),optional::enforced_data_loc
end
end interface
abstract interface
subroutine field_set_face(self,f,c_start,c_end,face,bc_start,bc_end,flow_rate_diff)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::f
real(8),intent(in)::c_start
real(8),intent(in)::c_end
integer(4),intent(in)::face
integer(4),intent(in),optional::bc_start
integer(4),intent(in),optional::bc_end
real(8),intent(in),optional::flow_rate_diff
end
end interface
abstract interface
subroutine field_set_face_from_field(self,f,f_start,c_end,face,bc_start,bc_end,flow_rate_diff)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::f
class(field_t),intent(in)::f_start
real(8),intent(in)::c_end
integer(4),intent(in)::face
integer(4),intent(in),optional::bc_start
integer(4),intent(in),optional::bc_end
real(8),intent(in),optional::flow_rate_diff
end
end interface
abstract interface
subroutine derive_field_from_gradients(self,field_out,dudx,dudy,dudz,dvdx,dvdy,dvdz,dwdx,dwdy,dwdz)
import::base_backend_t
import::field_t
class(base_backend_t)::self
class(field_t),intent(inout)::field_out
class(field_t),intent(in)::dudx
class(field_t),intent(in)::dudy
class(field_t),intent(in)::dudz
class(field_t),intent(in)::dvdx
class(field_t),intent(in)::dvdy
class(field_t),intent(in)::dvdz
class(field_t),intent(in)::dwdx
class(field_t),intent(in)::dwdy
class(field_t),intent(in)::dwdz
end
end interface
abstract interface
subroutine copy_data_to_f(self,f,data)
import::base_backend_t
import::field_t
class(base_backend_t),intent(inout)::self
class(field_t),intent(inout)::f
real(8),intent(in)::data(:,:,:)
end
end interface
abstract interface
subroutine copy_f_to_data(self,data,f)
import::base_backend_t
import::field_t
class(base_backend_t),intent(inout)::self
real(8),intent(out)::data(:,:,:)
class(field_t),intent(in)::f
end
end interface
abstract interface
subroutine alloc_tdsops(self,tdsops,n_tds,delta,operation,scheme,bc_start,bc_end,stretch,stretch_correct,n_halo,from_to,sym,c_nu,nu0_nu)
import::base_backend_t
import::tdsops_t
class(base_backend_t)::self
class(tdsops_t),allocatable,intent(inout)::tdsops
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
end
end interface
abstract interface
subroutine init_poisson_fft(self,mesh,xdirps,ydirps,zdirps,lowmem)
import::base_backend_t
import::dirps_t
import::mesh_t
class(base_backend_t)::self
type(mesh_t),intent(in)::mesh
type(dirps_t),intent(in)::xdirps
type(dirps_t),intent(in)::ydirps
type(dirps_t),intent(in)::zdirps
logical(4),intent(in),optional::lowmem
end
end interface
contains
subroutine base_init(self)
class(base_backend_t)::self
end
subroutine get_field_data(self,data,f,dir)
class(base_backend_t)::self
real(8),intent(out)::data(:,:,:)
class(field_t),intent(in)::f
integer(4),intent(in),optional::dir
end
subroutine set_field_data(self,f,data,dir)
class(base_backend_t)::self
class(field_t),intent(inout)::f
real(8),intent(in)::data(:,:,:)
integer(4),intent(in),optional::dir
end
end
